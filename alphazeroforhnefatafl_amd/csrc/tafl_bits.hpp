// tafl_bits.hpp — fixed-width board words as NL x 32-bit limbs (gfx950 VALU is 32-bit).
//
// Bit layout is the reference's: tile bit = row*ROW_WIDTH + col (game/bitfield.rs:72-74), with
// ROW_WIDTH 7 / 11 / 15 for 64 / 128 / 256-bit words (game/bitfield.rs:178-180).  The king nibble
// of the reference word (game/board/state.rs:127-147) is stripped at upload and re-attached at
// download; device words hold board bits only.
//
// Everything here is `__host__ __device__` so that tests/hostsim can compile the very same code
// with g++ and check it against the literal oracle on the CPU (no GPU in the build container).
// The product library never uses the host instantiation.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define TAFL_HD __host__ __device__ __forceinline__
#define TAFL_UNROLL _Pragma("unroll")
#else
#define TAFL_HD inline __attribute__((always_inline))
#define TAFL_UNROLL
#endif

namespace tafl {

// true if the predicate holds for ANY game of the wavefront (device) / for this game (host build): used to skip work that
// no game of the wave needs; results never depend on it.
TAFL_HD bool wave_any(bool p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __ballot(p) != 0ull;
#else
    return p;
#endif
}

template <int NL>
struct Bits {
    uint32_t w[NL];
};

template <int NL> TAFL_HD Bits<NL> bz() { Bits<NL> o; TAFL_UNROLL for (int i = 0; i < NL; ++i) o.w[i] = 0; return o; }
template <int NL> TAFL_HD Bits<NL> operator&(const Bits<NL>& a, const Bits<NL>& b) { Bits<NL> o; TAFL_UNROLL for (int i = 0; i < NL; ++i) o.w[i] = a.w[i] & b.w[i]; return o; }
template <int NL> TAFL_HD Bits<NL> operator|(const Bits<NL>& a, const Bits<NL>& b) { Bits<NL> o; TAFL_UNROLL for (int i = 0; i < NL; ++i) o.w[i] = a.w[i] | b.w[i]; return o; }
template <int NL> TAFL_HD Bits<NL> operator^(const Bits<NL>& a, const Bits<NL>& b) { Bits<NL> o; TAFL_UNROLL for (int i = 0; i < NL; ++i) o.w[i] = a.w[i] ^ b.w[i]; return o; }
template <int NL> TAFL_HD Bits<NL> operator~(const Bits<NL>& a) { Bits<NL> o; TAFL_UNROLL for (int i = 0; i < NL; ++i) o.w[i] = ~a.w[i]; return o; }
template <int NL> TAFL_HD Bits<NL> andn(const Bits<NL>& a, const Bits<NL>& b) { Bits<NL> o; TAFL_UNROLL for (int i = 0; i < NL; ++i) o.w[i] = a.w[i] & ~b.w[i]; return o; }
template <int NL> TAFL_HD Bits<NL>& operator|=(Bits<NL>& a, const Bits<NL>& b) { TAFL_UNROLL for (int i = 0; i < NL; ++i) a.w[i] |= b.w[i]; return a; }
template <int NL> TAFL_HD Bits<NL>& operator&=(Bits<NL>& a, const Bits<NL>& b) { TAFL_UNROLL for (int i = 0; i < NL; ++i) a.w[i] &= b.w[i]; return a; }
template <int NL> TAFL_HD bool any(const Bits<NL>& a) { uint32_t v = 0; TAFL_UNROLL for (int i = 0; i < NL; ++i) v |= a.w[i]; return v != 0; }
template <int NL> TAFL_HD bool eq(const Bits<NL>& a, const Bits<NL>& b) { uint32_t v = 0; TAFL_UNROLL for (int i = 0; i < NL; ++i) v |= a.w[i] ^ b.w[i]; return v == 0; }
template <int NL> TAFL_HD uint32_t popc(const Bits<NL>& a) { uint32_t c = 0; TAFL_UNROLL for (int i = 0; i < NL; ++i) c += (uint32_t)__builtin_popcount(a.w[i]); return c; }
// mask & (cond ? all : none)
template <int NL> TAFL_HD Bits<NL> gate(const Bits<NL>& a, bool cond) { uint32_t m = cond ? 0xFFFFFFFFu : 0u; Bits<NL> o; TAFL_UNROLL for (int i = 0; i < NL; ++i) o.w[i] = a.w[i] & m; return o; }
template <int NL> TAFL_HD Bits<NL> sel(bool cond, const Bits<NL>& a, const Bits<NL>& b) { Bits<NL> o; TAFL_UNROLL for (int i = 0; i < NL; ++i) o.w[i] = cond ? a.w[i] : b.w[i]; return o; }

// (cond ? a : b) by mask arithmetic: keeps constant operands as literals (a select of two constant objects is
// otherwise lowered to a pointer select + memory load)
template <int NL> TAFL_HD Bits<NL> blend(bool cond, const Bits<NL>& a, const Bits<NL>& b) {
    const uint32_t m = cond ? 0xFFFFFFFFu : 0u; Bits<NL> o;
    TAFL_UNROLL for (int i = 0; i < NL; ++i) o.w[i] = (a.w[i] & m) | (b.w[i] & ~m);
    return o;
}

// compile-time shifts (funnel shifts: v_alignbit_b32 on gfx950)
template <int K, int NL>
TAFL_HD Bits<NL> shl(const Bits<NL>& a) {
    constexpr int q = K / 32, r = K % 32;
    Bits<NL> o;
    TAFL_UNROLL
    for (int i = 0; i < NL; ++i) {
        const int s = i - q;
        uint32_t v = 0;
        if (s >= 0) {
            v = a.w[s] << r;
            if (r != 0 && s - 1 >= 0) v |= a.w[s - 1] >> ((32 - r) & 31);
        }
        o.w[i] = v;
    }
    return o;
}
template <int K, int NL>
TAFL_HD Bits<NL> shr(const Bits<NL>& a) {
    constexpr int q = K / 32, r = K % 32;
    Bits<NL> o;
    TAFL_UNROLL
    for (int i = 0; i < NL; ++i) {
        const int s = i + q;
        uint32_t v = 0;
        if (s < NL) {
            v = a.w[s] >> r;
            if (r != 0 && s + 1 < NL) v |= a.w[s + 1] << ((32 - r) & 31);
        }
        o.w[i] = v;
    }
    return o;
}

// run-time single-bit helpers.  No dynamic register indexing (that becomes scratch); built on 64-bit halves so that one
// v_lshl_b64 + a few selects replace a compare/select per 32-bit limb.
template <int NL> TAFL_HD Bits<NL> bit_at(uint32_t idx) {
    Bits<NL> o;
    const uint64_t b = 1ull << (idx & 63u);
    const uint32_t h = idx >> 6;
    TAFL_UNROLL for (int i = 0; i < NL / 2; ++i) {
        const uint64_t v = (h == (uint32_t)i) ? b : 0ull;
        o.w[2 * i] = (uint32_t)v; o.w[2 * i + 1] = (uint32_t)(v >> 32);
    }
    return o;
}
template <int NL> TAFL_HD bool test(const Bits<NL>& a, uint32_t idx) {
    const uint32_t h = idx >> 6; uint64_t v = 0;
    TAFL_UNROLL for (int i = 0; i < NL / 2; ++i) {
        const uint64_t x = (uint64_t)a.w[2 * i] | ((uint64_t)a.w[2 * i + 1] << 32);
        v |= (h == (uint32_t)i) ? x : 0ull;
    }
    return (v >> (idx & 63u)) & 1ull;
}
// index of lowest set bit (a must be non-zero)
template <int NL> TAFL_HD uint32_t lsb(const Bits<NL>& a) {
    uint32_t r = 0; bool found = false;
    TAFL_UNROLL for (int i = 0; i < NL; ++i) {
        const uint32_t v = a.w[i];
        if (!found && v) { r = (uint32_t)i * 32u + (uint32_t)__builtin_ctz(v); found = true; }
    }
    return r;
}
// index of highest set bit (a must be non-zero)
template <int NL> TAFL_HD uint32_t msb(const Bits<NL>& a) {
    uint32_t r = 0; bool found = false;
    TAFL_UNROLL for (int i = NL - 1; i >= 0; --i) {
        const uint32_t v = a.w[i];
        if (!found && v) { r = (uint32_t)i * 32u + 31u - (uint32_t)__builtin_clz(v); found = true; }
    }
    return r;
}
// all bits strictly below idx (idx <= NL*32)
template <int NL> TAFL_HD Bits<NL> below(uint32_t idx) {
    Bits<NL> o;
    const uint64_t m = (1ull << (idx & 63u)) - 1ull;
    const uint32_t h = idx >> 6;
    TAFL_UNROLL for (int i = 0; i < NL / 2; ++i) {
        const uint64_t v = ((uint32_t)i < h) ? ~0ull : (((uint32_t)i == h) ? m : 0ull);
        o.w[2 * i] = (uint32_t)v; o.w[2 * i + 1] = (uint32_t)(v >> 32);
    }
    return o;
}
// 5-bit window around idx: bit k of the result = bit (idx - 2 + k) of a (bits outside the word read as 0)
template <int NL> TAFL_HD uint32_t window5(const Bits<NL>& a, uint32_t idx) {
    const Bits<NL> b = shl<2>(a);                      // bit (idx-2) of a is bit idx of b: no negative positions
    const uint32_t wi = idx >> 5, off = idx & 31;
    uint32_t lo = 0, hi = 0;
    TAFL_UNROLL for (int i = 0; i < NL; ++i) {         // limb wi and its successor (selects, no dynamic indexing)
        lo = ((uint32_t)i == wi) ? b.w[i] : lo;
        hi = ((uint32_t)i == wi + 1) ? b.w[i] : hi;
    }
    // funnel shift of the limb pair (v_alignbit_b32 on gfx950)
    const uint32_t v = off ? ((lo >> off) | (hi << ((32 - off) & 31))) : lo;
    return v & 31u;
}

// position of the j-th (0-based) set bit of a 32-bit word, j < popcount(v)
TAFL_HD uint32_t nth_set_bit32(uint32_t v, uint32_t j) {
    uint32_t pos = 0;
    TAFL_UNROLL
    for (int width = 16; width >= 1; width >>= 1) {
        const uint32_t lowmask = ((1u << width) - 1u) << pos;
        const uint32_t c = (uint32_t)__builtin_popcount(v & lowmask);
        if (j >= c) { j -= c; pos += (uint32_t)width; }
    }
    return pos;
}
// index of the j-th (0-based) set bit, j < popc(a).  Mask arithmetic only: a select chain here gets turned into a
// scratch-backed indexed load by the compiler.
template <int NL> TAFL_HD uint32_t nth_set_bit(const Bits<NL>& a, uint32_t j) {
    uint32_t word = 0, base = 0, rem = 0, acc = 0;
    TAFL_UNROLL for (int i = 0; i < NL; ++i) {
        const uint32_t c = (uint32_t)__builtin_popcount(a.w[i]);
        const uint32_t hit = (j >= acc && j < acc + c) ? 0xFFFFFFFFu : 0u;      // at most one word hits
        word |= a.w[i] & hit; base |= ((uint32_t)i * 32u) & hit; rem |= (j - acc) & hit;
        acc += c;
    }
    return base + nth_set_bit32(word, rem);
}

}  // namespace tafl
