// The fused image kernel rounds with q = mulhi(2 acc + D, magic) >> shift (ucfp_amd/csrc/any_magic.h): check the magic
// numbers against plain division at every quotient boundary, for every den = 2 w h of a grid of geometries plus random ones.
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "../../ucfp_amd/csrc/any_magic.h"

static uint64_t checked = 0;
static int check_den(uint32_t den) {
    uint32_t m = 0, s = 0;
    if (!ucfp::any_magic(den, &m, &s)) return den >= (1u << 23) || den < 2 ? 0 : 1;
    for (uint64_t k = 0; k <= 256; k++)
        for (int d = -2; d <= 2; d++) {
            const int64_t num = (int64_t)(k * den) + d;
            if (num < 0 || num >= (int64_t)256 * den) continue;
            const uint32_t q = (uint32_t)(((uint64_t)(uint32_t)num * m) >> 32) >> s;
            checked++;
            if (q != (uint32_t)(num / den)) {
                std::printf("den %u num %lld: got %u want %u\n", den, (long long)num, q, (uint32_t)(num / den));
                return 1;
            }
        }
    // the largest numerator the kernel forms: 2 * 255 D + D = 511 D, D = den / 2
    const uint64_t top = 511ull * (den / 2);
    if (top >> 32) return 1;
    const uint32_t q = (uint32_t)((top * m) >> 32) >> s;
    return q != (uint32_t)(top / den);
}

int main() {
    int bad = 0;
    for (uint32_t w = 1; w <= 2048 && !bad; w += (w < 64 ? 1 : 37))
        for (uint32_t h = 1; h <= 8192 && !bad; h += (h < 64 ? 1 : 53)) {
            const uint64_t d = 2ull * w * h;
            if (d < (1u << 23)) bad |= check_den((uint32_t)d);
        }
    uint64_t x = 88172645463325252ull;
    for (int i = 0; i < 200000 && !bad; i++) {
        x ^= x << 13, x ^= x >> 7, x ^= x << 17;
        bad |= check_den(2u + (uint32_t)(x % ((1u << 23) - 2)));
    }
    for (uint32_t den = (1u << 23) - 64; den < (1u << 23) && !bad; den++) bad |= check_den(den);
    for (uint32_t den = 2; den < 5000 && !bad; den++) bad |= check_den(den);
    std::printf("%s: %llu quotients checked\n", bad ? "FAILED" : "ok", (unsigned long long)checked);
    return bad;
}
