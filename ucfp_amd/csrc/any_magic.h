// any_magic.h -- the rounding division of the fused image kernel as one multiply-high (host side: plain C++).
#pragma once
#include <stdint.h>

namespace ucfp {

// floor(num / den) = mulhi(num, magic) >> shift for every num < 256 den (the rounding division's range: the quotient is a
// pixel).  With 2^k >= 256 den^2 and magic = ceil(2^k / den):  0 <= magic den - 2^k < den <= 2^k / (256 den), which is the
// exactness condition of division by an invariant multiplier for numerators below 256 den.  den < 2^23 keeps the magic
// number below 2^32; k >= 32 so that the product's high word, shifted, is the quotient.
inline bool any_magic(uint32_t den, uint32_t* magic, uint32_t* shift) {
    if (den < 2 || den >= (1u << 23)) return false;
    uint32_t k = 32;
    while (((unsigned __int128)1 << k) < (unsigned __int128)256 * den * den) k++;
    const unsigned __int128 m = (((unsigned __int128)1 << k) + den - 1) / den;
    if (m >> 32) return false;
    *magic = (uint32_t)m;
    *shift = k - 32;
    return true;
}

}  // namespace ucfp
