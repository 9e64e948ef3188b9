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

// Section timers for profiling builds only (-DTAFL_PROF): per-section shader-clock totals, one lane per wave reporting.
// Marks must sit on the path every game takes every ply.  Product builds compile them away.
#if defined(TAFL_PROF) && defined(__HIP_DEVICE_COMPILE__)
extern "C" __device__ unsigned long long tafl_prof_acc[4096 * 32];   // one row per workgroup: no contention
#define TAFL_PROF_ADD_(k, v) do { const unsigned long long b_ = __ballot(1); \
    if ((int)__lane_id() == __ffsll((long long)b_) - 1) atomicAdd(&tafl_prof_acc[(blockIdx.x & 4095u) * 32u + (k)], (unsigned long long)(v)); } while (0)
#define TAFL_PROF_BEGIN(k) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_readcyclecounter(); TAFL_PROF_ADD_(k, 0ull - t_); __builtin_amdgcn_sched_barrier(0); } while (0)
#define TAFL_PROF_END(k) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_readcyclecounter(); TAFL_PROF_ADD_(k, t_); __builtin_amdgcn_sched_barrier(0); } while (0)
#define TAFL_PROF_SPLIT(k) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_readcyclecounter(); TAFL_PROF_ADD_(k, t_); TAFL_PROF_ADD_((k) + 1, 0ull - t_); __builtin_amdgcn_sched_barrier(0); } while (0)
#define TAFL_PROF_COUNT(k) TAFL_PROF_ADD_(k, 1ull)
#else
#define TAFL_PROF_BEGIN(k) do {} while (0)
#define TAFL_PROF_END(k) do {} while (0)
#define TAFL_PROF_SPLIT(k) do {} while (0)
#define TAFL_PROF_COUNT(k) do {} while (0)
#endif

// event counters for host-side statistics builds only (-DTAFL_STAT, tests/hostsim experiments): how often a game takes a path
#if defined(TAFL_STAT) && !defined(__HIP_DEVICE_COMPILE__)
extern "C" unsigned long long tafl_stat_acc[32];
#define TAFL_STAT_HIT(k) (++tafl_stat_acc[(k)])
#else
#define TAFL_STAT_HIT(k) ((void)0)
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

// tile-index arithmetic with 24-bit multiplies (v_mul_u32_u24 / v_mad_u32_u24 are full rate on gfx950; the 32-bit
// v_mul_lo_u32 / v_mul_hi_u32 a division by a constant compiles to are quarter rate).  Operands are tile indices (< 4694).
TAFL_HD uint32_t mul24(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul24(a, b);
#else
    return (a & 0xFFFFFFu) * (b & 0xFFFFFFu);
#endif
}
template <int W> TAFL_HD uint32_t div_w(uint32_t x) {          // x / W, exact for x < 4694 (W in {7, 11, 13, 15})
    static_assert(W == 7 || W == 11 || W == 13 || W == 15, "div_w: reciprocal checked for these row widths only");
    return mul24(x, 65536u / (uint32_t)W + 1u) >> 16;
}
template <int W> TAFL_HD uint32_t mod_w(uint32_t x) { return x - mul24(div_w<W>(x), (uint32_t)W); }

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
// 64-bit field of `a` that starts BACK (<= 32) bits below idx: bit k of the result = bit (idx - BACK + k) of a; positions
// outside the word read as 0.  With BACK = 2*W the tile idx sits at bit 2W and its row / column neighbours at distance 1
// and 2 sit at the fixed bits 2W-2..2W+2 and 0, W, 3W, 4W (4W + 1 <= 61 for W <= 15).
template <int BACK, int NL> TAFL_HD uint64_t field64(const Bits<NL>& a, uint32_t idx) {
    static_assert(BACK >= 0 && BACK <= 32, "field64: BACK out of range");
    if constexpr (NL == 2) {                           // the whole word is one 64-bit value: a shift either way
        const uint64_t v = (uint64_t)a.w[0] | ((uint64_t)a.w[1] << 32);
        return idx >= (uint32_t)BACK ? (v >> ((idx - (uint32_t)BACK) & 63u)) : (v << (((uint32_t)BACK - idx) & 63u));
    } else {
        const uint32_t p = idx + 32u - (uint32_t)BACK; // position in [zero limb, a.w[0], a.w[1], ...]
        const uint32_t wi = p >> 5, off = p & 31u;
        uint32_t x0 = 0, x1 = 0, x2 = 0;
        TAFL_UNROLL for (int i = 0; i < NL; ++i) {     // limbs wi-1, wi, wi+1 of a (selects, no dynamic indexing)
            x0 = ((uint32_t)i + 1u == wi) ? a.w[i] : x0;
            x1 = ((uint32_t)i == wi) ? a.w[i] : x1;
            x2 = ((uint32_t)i == wi + 1u) ? a.w[i] : x2;
        }
        // funnel shifts (v_alignbit_b32 on gfx950)
        const uint32_t lo = (uint32_t)((((uint64_t)x1 << 32) | x0) >> off);
        const uint32_t hi = (uint32_t)((((uint64_t)x2 << 32) | x1) >> off);
        return (uint64_t)lo | ((uint64_t)hi << 32);
    }
}
// inverse of field64: bit k of f goes to bit (idx - BACK + k); bits that fall outside the word are dropped
template <int BACK, int NL> TAFL_HD Bits<NL> deposit64(uint64_t f, uint32_t idx) {
    static_assert(BACK >= 0 && BACK <= 32, "deposit64: BACK out of range");
    Bits<NL> o;
    if constexpr (NL == 2) {
        const uint64_t v = idx >= (uint32_t)BACK ? (f << ((idx - (uint32_t)BACK) & 63u)) : (f >> (((uint32_t)BACK - idx) & 63u));
        o.w[0] = (uint32_t)v; o.w[1] = (uint32_t)(v >> 32);
    } else {
        const uint32_t p = idx + 32u - (uint32_t)BACK;
        const uint32_t wi = p >> 5, off = p & 31u;
        const uint32_t flo = (uint32_t)f, fhi = (uint32_t)(f >> 32);
        const uint32_t y0 = flo << off;
        const uint32_t y1 = (uint32_t)((f << off) >> 32);
        const uint32_t y2 = (uint32_t)(((uint64_t)fhi << off) >> 32);
        TAFL_UNROLL for (int i = 0; i < NL; ++i)
            o.w[i] = ((uint32_t)i + 1u == wi) ? y0 : ((uint32_t)i == wi) ? y1 : ((uint32_t)i == wi + 1u) ? y2 : 0u;
    }
    return o;
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
