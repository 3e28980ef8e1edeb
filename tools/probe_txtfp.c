/* Bounded probe for txtfp 0.2.0's MinHash slot-0 derivation (VERDICT r1 "next" 1a).
 *
 * Pin: MinHash-128 of "the quick brown fox jumps over the lazy dog" has slot 0 = 0x06818a8cc8cc6aa2
 * (/root/reference/src/server/tests.rs:1153-1157).  txtfp's source is absent; its Cargo.lock
 * dependency list has xxhash-rust, wide, blake3, ahash and NO rand crate, so per-slot parameters must
 * come from a hand-rolled generator.  This program enumerates such constructions and prints any hit.
 * Build: gcc -O3 -fopenmp -I<dir holding xxhash.h> tools/probe_txtfp.c -o /tmp/probe_txtfp
 * Result of the round-2 run is recorded in DESIGN.md section 2.
 */
#define XXH_INLINE_ALL
#include "xxhash.h"
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

static const uint64_t TARGET = 0x06818a8cc8cc6aa2ull;
static const uint64_t P61 = (1ull << 61) - 1;
static const char* WORDS[9] = {"the", "quick", "brown", "fox", "jumps", "over", "the", "lazy", "dog"};

typedef struct { uint8_t b[128]; int n; } shingle;
#define MAXS 64
typedef struct { shingle s[MAXS]; int n; char name[64]; } enc;
static enc ENC[64];
static int NENC = 0;

static void add_join(const char* sep, int seplen, int k, int trailing, const char* name) {
    enc* e = &ENC[NENC++];
    snprintf(e->name, sizeof e->name, "%s/k%d", name, k);
    e->n = 0;
    for (int i = 0; i + k <= 9; i++) {
        shingle* s = &e->s[e->n++];
        s->n = 0;
        for (int j = 0; j < k; j++) {
            int l = (int)strlen(WORDS[i + j]);
            memcpy(s->b + s->n, WORDS[i + j], l); s->n += l;
            if (j + 1 < k || trailing) { memcpy(s->b + s->n, sep, seplen); s->n += seplen; }
        }
    }
}
static void add_lenprefix(int width, int k) {
    enc* e = &ENC[NENC++];
    snprintf(e->name, sizeof e->name, "lenprefix%d/k%d", width, k);
    e->n = 0;
    for (int i = 0; i + k <= 9; i++) {
        shingle* s = &e->s[e->n++];
        s->n = 0;
        for (int j = 0; j < k; j++) {
            uint64_t l = strlen(WORDS[i + j]);
            memcpy(s->b + s->n, &l, width); s->n += width;
            memcpy(s->b + s->n, WORDS[i + j], l); s->n += (int)l;
        }
    }
}
/* tokens = UAX#29 split_word_bounds (words AND the spaces between them): 17 tokens */
static void add_bounds(int k) {
    const char* toks[17]; int nt = 0;
    for (int i = 0; i < 9; i++) { toks[nt++] = WORDS[i]; if (i < 8) toks[nt++] = " "; }
    enc* e = &ENC[NENC++];
    snprintf(e->name, sizeof e->name, "bounds/k%d", k);
    e->n = 0;
    for (int i = 0; i + k <= nt; i++) {
        shingle* s = &e->s[e->n++]; s->n = 0;
        for (int j = 0; j < k; j++) { int l = (int)strlen(toks[i + j]); memcpy(s->b + s->n, toks[i + j], l); s->n += l; }
    }
}
/* shingle = concatenation of the k token hashes (8 B LE each) */
static void add_tokhash(int k, uint64_t seed) {
    enc* e = &ENC[NENC++];
    snprintf(e->name, sizeof e->name, "tokhash(seed%llu)/k%d", (unsigned long long)seed, k);
    e->n = 0;
    for (int i = 0; i + k <= 9; i++) {
        shingle* s = &e->s[e->n++]; s->n = 0;
        for (int j = 0; j < k; j++) {
            uint64_t h = XXH3_64bits_withSeed(WORDS[i + j], strlen(WORDS[i + j]), seed);
            memcpy(s->b + s->n, &h, 8); s->n += 8;
        }
    }
}
/* character k-grams of the canonical text */
static void add_chars(int k) {
    const char* t = "the quick brown fox jumps over the lazy dog";
    int n = (int)strlen(t);
    enc* e = &ENC[NENC++];
    snprintf(e->name, sizeof e->name, "chars/k%d", k);
    e->n = 0;
    for (int i = 0; i + k <= n && e->n < MAXS; i++) { shingle* s = &e->s[e->n++]; memcpy(s->b, t + i, k); s->n = k; }
}

static inline uint64_t sm_next(uint64_t* st) {
    uint64_t z = (*st += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static inline uint64_t sm_mix(uint64_t z) { uint64_t s = z - 0x9E3779B97F4A7C15ull; return sm_next(&s); }
static inline uint64_t fmix64(uint64_t k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33; return k;
}
static inline uint64_t xs64star(uint64_t* s) { uint64_t x = *s; x ^= x >> 12; x ^= x << 25; x ^= x >> 27; *s = x; return x * 0x2545F4914F6CDD1Dull; }
static inline uint64_t wymix(uint64_t a, uint64_t b) { __uint128_t r = (__uint128_t)a * b; return (uint64_t)r ^ (uint64_t)(r >> 64); }
static inline uint64_t mod61(__uint128_t x) {
    uint64_t lo = (uint64_t)(x & P61), hi = (uint64_t)(x >> 61);
    uint64_t r = lo + (hi & P61) + (uint64_t)(x >> 122);
    while (r >= P61) r -= P61;
    return r;
}

static void hit(const char* fam, const char* encname, unsigned long long param, int variant) {
    printf("HIT family=%s enc=%s param=%llu variant=%d\n", fam, encname, param, variant);
    fflush(stdout);
}

int main(int argc, char** argv) {
    uint64_t NS = argc > 1 ? strtoull(argv[1], 0, 0) : (1ull << 26);   /* seeds per family */
    const char* seps[] = {" ", "", "\0", "\x1f", "_", "\n", ",", "\t", "|", "-", "\x1e", "+"};
    const int seplen[] = {1, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};
    const char* sepname[] = {"sp", "none", "nul", "us", "under", "nl", "comma", "tab", "bar", "dash", "rs", "plus"};
    for (int i = 0; i < 12; i++) add_join(seps[i], seplen[i], 5, 0, sepname[i]);
    add_join(" ", 1, 5, 1, "sp-trailing");
    for (int w = 1; w <= 8; w *= 2) add_lenprefix(w, 5);
    add_bounds(5); add_bounds(9); add_bounds(10);
    add_tokhash(5, 0);
    add_join(" ", 1, 4, 0, "sp"); add_join(" ", 1, 6, 0, "sp"); add_join(" ", 1, 3, 0, "sp"); add_join(" ", 1, 1, 0, "sp");
    add_chars(5);
    printf("%d encodings, %llu seeds per family\n", NENC, (unsigned long long)NS);

    /* base hashes per (encoding, shingle): xxh3_64 seed 0, xxh3_128 lo/hi, xxh64 seed 0 */
    static uint64_t H[64][MAXS][4];
    for (int e = 0; e < NENC; e++)
        for (int s = 0; s < ENC[e].n; s++) {
            shingle* sh = &ENC[e].s[s];
            H[e][s][0] = XXH3_64bits(sh->b, sh->n);
            XXH128_hash_t h128 = XXH3_128bits(sh->b, sh->n);
            H[e][s][1] = h128.low64; H[e][s][2] = h128.high64;
            H[e][s][3] = XXH64(sh->b, sh->n, 0);
        }

    /* F0: fixed (seed-free) transforms of the base hash, min and max */
    for (int e = 0; e < NENC; e++)
        for (int b = 0; b < 4; b++) {
            uint64_t mn[12], mx[12];
            for (int v = 0; v < 12; v++) { mn[v] = ~0ull; mx[v] = 0; }
            for (int s = 0; s < ENC[e].n; s++) {
                uint64_t h = H[e][s][b], v[12];
                v[0] = h; v[1] = sm_mix(h); v[2] = fmix64(h); v[3] = h * 0x9E3779B97F4A7C15ull;
                v[4] = h & P61; v[5] = h % P61; v[6] = h >> 3; v[7] = __builtin_bswap64(h);
                v[8] = sm_mix(h) % P61; v[9] = XXH3_64bits(&h, 8); v[10] = h ^ (h >> 32); v[11] = (h << 32) | (h >> 32);
                for (int k = 0; k < 12; k++) { if (v[k] < mn[k]) mn[k] = v[k]; if (v[k] > mx[k]) mx[k] = v[k]; }
            }
            for (int k = 0; k < 12; k++) { if (mn[k] == TARGET) hit("F0min", ENC[e].name, b, k); if (mx[k] == TARGET) hit("F0max", ENC[e].name, b, k); }
        }
    /* F0b: double hashing with 1-based slot index (slot0 = h1 + h2) from the 128-bit halves or two seeds */
    for (int e = 0; e < NENC; e++) {
        uint64_t mn[4] = {~0ull, ~0ull, ~0ull, ~0ull};
        for (int s = 0; s < ENC[e].n; s++) {
            shingle* sh = &ENC[e].s[s];
            uint64_t a = H[e][s][1], b = H[e][s][2], c = H[e][s][0], d = XXH3_64bits_withSeed(sh->b, sh->n, 1);
            uint64_t v[4] = {a + b, b + a * 1, c + d, c + (d | 1)};
            for (int k = 0; k < 4; k++) if (v[k] < mn[k]) mn[k] = v[k];
        }
        for (int k = 0; k < 4; k++) if (mn[k] == TARGET) hit("F0b", ENC[e].name, 0, k);
    }
    printf("F0 done\n"); fflush(stdout);

    /* F1: xxh3_64_withSeed(shingle, seed) with seed = g(S): identity, splitmix stream, splitmix mix, fmix, golden multiples,
     *     xorshift64*, wyhash-style; also XXH64 with those seeds.  S in [0, NS). */
    #pragma omp parallel for schedule(dynamic, 4096)
    for (uint64_t S = 0; S < NS; S++) {
        uint64_t st = S, xs = S ? S : 1;
        uint64_t seeds[8] = {S, sm_next(&st), sm_next(&st), fmix64(S), S * 0x9E3779B97F4A7C15ull, xs64star(&xs),
                             wymix(S ^ 0xa0761d6478bd642full, 0xe7037ed1a0b428dbull), S ^ 0x9E3779B97F4A7C15ull};
        int ne = S < (1u << 20) ? NENC : 3;          /* all encodings for small S, the 3 likeliest beyond */
        for (int v = 0; v < 8; v++)
            for (int e = 0; e < ne; e++) {
                uint64_t mn = ~0ull, mn2 = ~0ull;
                for (int s = 0; s < ENC[e].n; s++) {
                    uint64_t h = XXH3_64bits_withSeed(ENC[e].s[s].b, ENC[e].s[s].n, seeds[v]);
                    if (h < mn) mn = h;
                    if (v < 2 && S < (1u << 22)) { uint64_t g = XXH64(ENC[e].s[s].b, ENC[e].s[s].n, seeds[v]); if (g < mn2) mn2 = g; }
                }
                if (mn == TARGET) hit("F1-xxh3-seeded", ENC[e].name, S, v);
                if (mn2 == TARGET) hit("F1-xxh64-seeded", ENC[e].name, S, v);
            }
    }
    printf("F1 done\n"); fflush(stdout);

    /* F2: permutation of the seed-0 base hash with (a, b) from a generator seeded S:
     *     wrapping a*h+b, (a*h+b) mod 2^61-1 (h reduced or not), h^a, sm_mix(h^a), sm_mix(h+a), fmix64(h^a), wymix(h^a, b) */
    uint64_t NS2 = argc > 2 ? strtoull(argv[2], 0, 0) : (1ull << 20);
    #pragma omp parallel for schedule(dynamic, 4096)
    for (uint64_t S = 0; S < NS2; S++) {
        uint64_t st = S, xs = S ? S : 1;
        uint64_t g[4][2];
        g[0][0] = sm_next(&st); g[0][1] = sm_next(&st);
        g[1][0] = xs64star(&xs); g[1][1] = xs64star(&xs);
        g[2][0] = fmix64(2 * S + 1); g[2][1] = fmix64(2 * S + 2);
        g[3][0] = sm_mix(2 * S); g[3][1] = sm_mix(2 * S + 1);
        for (int gi = 0; gi < 4; gi++)
            for (int sw = 0; sw < 2; sw++) {
                uint64_t a = g[gi][sw], b = g[gi][1 - sw];
                uint64_t a61 = a % P61, b61 = b % P61; if (!a61) a61 = 1;
                for (int e = 0; e < NENC; e++)
                    for (int bh = 0; bh < 3; bh++) {
                        uint64_t mn[10];
                        for (int k = 0; k < 10; k++) mn[k] = ~0ull;
                        for (int s = 0; s < ENC[e].n; s++) {
                            uint64_t h = H[e][s][bh], v[10];
                            v[0] = a * h + b; v[1] = (a | 1) * h + b;
                            v[2] = mod61((__uint128_t)a61 * (h % P61) + b61);
                            v[3] = mod61((__uint128_t)a61 * h + b61);
                            v[4] = h ^ a; v[5] = sm_mix(h ^ a); v[6] = sm_mix(h + a); v[7] = fmix64(h ^ a);
                            v[8] = wymix(h ^ a, b); v[9] = mod61((__uint128_t)a61 * (h & 0xffffffffull) + b61);
                            for (int k = 0; k < 10; k++) if (v[k] < mn[k]) mn[k] = v[k];
                        }
                        for (int k = 0; k < 10; k++) if (mn[k] == TARGET) hit("F2-perm", ENC[e].name, S, gi * 1000 + sw * 100 + bh * 10 + k);
                    }
            }
    }
    printf("F2 done\n"); fflush(stdout);
    return 0;
}
