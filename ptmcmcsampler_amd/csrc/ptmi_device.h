// ptmi_device.h -- device-side building blocks of libptmi (gfx950 only).
//
// Everything a parity decision depends on is spelled with IEEE-exact operations
// (+ - * / sqrt, explicit fma) in a fixed order, so the kernels agree bit for bit
// with the CPU oracle used by the tests.  Compile with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ptmi_tables.h"

namespace ptmi {

typedef unsigned long long u64;
typedef unsigned int u32;
typedef double ptmi_dev_d2 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------- cross-lane
// DPP moves: full-rate lane permutations inside a row of 16 lanes (no LDS traffic).
// (every control used here reads a valid lane for every lane, so no "old" value is needed: with update_dpp(0, ...) hipcc
// emitted a v_mov_b32 0 in front of every DPP move)
template <int CTRL>
__device__ __forceinline__ u32 dpp32(u32 v)
{
    return (u32)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xF, 0xF, true);
}
template <int CTRL>
__device__ __forceinline__ u64 dpp64(u64 v)
{
    u32 lo = dpp32<CTRL>((u32)v), hi = dpp32<CTRL>((u32)(v >> 32));
    return ((u64)hi << 32) | lo;
}
template <int CTRL>
__device__ __forceinline__ double dppf64(double v)
{
    return __longlong_as_double((long long)dpp64<CTRL>((u64)__double_as_longlong(v)));
}
// broadcast lane J of every quad to the quad
template <int J>
__device__ __forceinline__ u64 quad_bcast(u64 v) { return dpp64<J * 0x55>(v); }
template <int J>
__device__ __forceinline__ double quad_bcastf(double v) { return dppf64<J * 0x55>(v); }

// All-reduce (sum) over the G lanes that share a chain.  Pairing order is the xor
// butterfly m = G/2 .. 1; after the m = 8 step the data has period 8, so a row rotate
// by 4 pairs exactly the lanes l and l^4.
// p[l] + p[l ^ 32] and p[l] + p[l ^ 16] without the LDS round trip of a shuffle: v_permlane32_swap trades the upper half of its first
// operand for the lower half of its second (v_permlane16_swap: the odd rows of 16 lanes for the even rows), so a register swapped
// with a copy of itself leaves the lower halves (even rows) in the first result and the upper halves (odd rows) in the second on
// every lane; their sum is the pair's sum -- for the upper lane with its operands the other way round, which a sum does not see.
__device__ __forceinline__ double sum_xor32(double p)
{
    const u64 b = (u64)__double_as_longlong(p);
    const auto lo = __builtin_amdgcn_permlane32_swap((u32)b, (u32)b, false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((u32)(b >> 32), (u32)(b >> 32), false, false);
    return __longlong_as_double((long long)(((u64)hi[0] << 32) | lo[0])) + __longlong_as_double((long long)(((u64)hi[1] << 32) | lo[1]));
}
__device__ __forceinline__ double sum_xor16(double p)
{
    const u64 b = (u64)__double_as_longlong(p);
    const auto lo = __builtin_amdgcn_permlane16_swap((u32)b, (u32)b, false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((u32)(b >> 32), (u32)(b >> 32), false, false);
    return __longlong_as_double((long long)(((u64)hi[0] << 32) | lo[0])) + __longlong_as_double((long long)(((u64)hi[1] << 32) | lo[1]));
}
template <int G>
__device__ __forceinline__ double group_sum(double p)
{
    if (G >= 64) p = sum_xor32(p);
    if (G >= 32) p = sum_xor16(p);
    if (G >= 16) p = p + dppf64<0x128>(p);   // row_ror:8  == xor 8
    if (G >= 8) p = p + dppf64<0x124>(p);    // row_ror:4  == xor 4 on period-8 data
    if (G >= 4) p = p + dppf64<0x4E>(p);     // quad_perm [2,3,0,1] == xor 2
    if (G >= 2) p = p + dppf64<0xB1>(p);     // quad_perm [1,0,3,2] == xor 1
    return p;
}
// true iff `ok` holds on all G lanes of the caller's group
template <int G>
__device__ __forceinline__ bool group_all(bool ok)
{
    const u64 m = __ballot(ok);
    const int lane = (int)(threadIdx.x & 63);
    const u64 full = G == 64 ? ~0ull : ((1ull << G) - 1ull);
    return ((m >> (lane & ~(G - 1))) & full) == full;
}

// --------------------------------------------------------------------- Philox
// Philox4x32-10 (Salmon et al., SC'11).  ctr = (iter_lo, iter_hi, stream, slot), key = seed.
// The generator in three pieces so that a caller can place other work between the rounds (am_mfma_product).
struct PhiloxState { u32 c0, c1, c2, c3, k0, k1; };
__device__ __forceinline__ void philox_begin(PhiloxState &p, u64 seed, u64 iter, u32 stream, u32 slot)
{
    p.c0 = (u32)iter; p.c1 = (u32)(iter >> 32); p.c2 = stream; p.c3 = slot;
    p.k0 = (u32)seed; p.k1 = (u32)(seed >> 32);
}
__device__ __forceinline__ void philox_round(PhiloxState &p)
{
    // one v_mad_u64_u32 per product (hi and lo at once).  Left to itself hipcc splits each into v_mul_hi_u32 + v_mul_lo_u32:
    // four multiplies per round instead of two (tools/inst_rates.hip: 23 against 13.4 cycles per round and SIMD)
    u64 p0, p1, cy0, cy1;
    asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(p0), "=s"(cy0) : "v"(p.c0), "s"(0xD2511F53u));
    asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(p1), "=s"(cy1) : "v"(p.c2), "s"(0xCD9E8D57u));
    const u32 h0 = (u32)(p0 >> 32), l0 = (u32)p0, h1 = (u32)(p1 >> 32), l1 = (u32)p1;
    // three-input xor in one instruction (v_bitop3_b32, truth table 0x96): hipcc emits two v_xor_b32 for a ^ b ^ c
    const u32 n0 = __builtin_amdgcn_bitop3_b32(h1, p.c1, p.k0, 0x96), n2 = __builtin_amdgcn_bitop3_b32(h0, p.c3, p.k1, 0x96);
    p.c0 = n0; p.c1 = l1; p.c2 = n2; p.c3 = l0;
    p.k0 += 0x9E3779B9u; p.k1 += 0xBB67AE85u;
}
__device__ __forceinline__ void philox_end(const PhiloxState &p, u64 &w0, u64 &w1)
{
    w0 = ((u64)p.c1 << 32) | p.c0;
    w1 = ((u64)p.c3 << 32) | p.c2;
}
__device__ __forceinline__ void philox_words(u64 seed, u64 iter, u32 stream, u32 slot, u64 &w0, u64 &w1)
{
    PhiloxState p;
    philox_begin(p, seed, iter, stream, slot);
#pragma unroll
    for (int r = 0; r < 10; ++r) philox_round(p);
    philox_end(p, w0, w1);
}
enum : u32 { SLOT_A = 0, SLOT_B = 1, SLOT_C = 2, SLOT_D = 3, SLOT_SWAP = 0x10000u, SLOT_AM = 0x1000000u };

__device__ __forceinline__ double w2uniform(u64 w) { return (double)(w >> 11) * 0x1.0p-53; }            // [0,1)
__device__ __forceinline__ double w2uniform_open(u64 w) { return (double)((w >> 11) + 1ull) * 0x1.0p-53; } // (0,1]
__device__ __forceinline__ u64 w2index(u64 w, u64 n) { return __umul64hi(w, n); }

// ------------------------------------------------------- deterministic libm
__device__ __forceinline__ double det_log(double x)
{
    if (x != x) return x;
    if (x <= 0.0) return x == 0.0 ? -__builtin_inf() : __builtin_nan("");
    if (x == __builtin_inf()) return x;
    u64 u = (u64)__double_as_longlong(x);
    int k = 0;
    if ((u >> 52) == 0) { x *= 0x1.0p54; u = (u64)__double_as_longlong(x); k = -54; }
    k += (int)(u >> 52) - 1023;
    const u64 man = u & 0x000FFFFFFFFFFFFFull;
    if (man >= 0x6A09E667F3BCDull) { k += 1; u = man | 0x3FE0000000000000ull; }
    else u = man | 0x3FF0000000000000ull;
    const double f = __longlong_as_double((long long)u) - 1.0;
    const double s = f / (2.0 + f);
    const double z = s * s, w = z * z;
    const double t1 = w * (3.999999999940941908e-01 + w * (2.222219843214978396e-01 + w * 1.531383769920937332e-01));
    const double t2 = z * (6.666666666666735130e-01 + w * (2.857142874366239149e-01 + w * (1.818357216161805012e-01 + w * 1.479819860511658591e-01)));
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)k;
    return dk * 6.93147180369123816490e-01 - ((hfsq - (s * (hfsq + R) + dk * 1.90821492927058770002e-10)) - f);
}

// Straight-line: the special cases are selected at the end (a chain per wave lane that branches on them costs more
// in exec-mask bookkeeping than the few extra instructions); same results as the oracle's early returns.
__device__ __forceinline__ double det_exp(double xin)
{
    const bool isnan = xin != xin, over = xin > 7.09782712893383973096e+02, under = xin < -7.45133219101941108420e+02;
    const double x = (isnan || over || under) ? 0.0 : xin;           // the main path runs on a harmless argument then
    const int k = (int)(1.44269504088896338700e+00 * x + (x < 0.0 ? -0.5 : 0.5));
    const double dk = (double)k;
    const double hi = x - dk * 6.93147180369123816490e-01;
    const double lo = dk * 1.90821492927058770002e-10;
    const double r = hi - lo;
    const double t = r * r;
    const double c = r - t * (1.66666666666666019037e-01 + t * (-2.77777777770155933842e-03 + t * (6.61375632143793436117e-05 +
                     t * (-1.65339022054652515390e-06 + t * 4.13813679705723846039e-08))));
    const double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
    const u64 yb = (u64)__double_as_longlong(y);
    const double normal = __longlong_as_double((long long)(yb + ((u64)(long long)k << 52)));                 // k = 0: y itself
    const double tiny = __longlong_as_double((long long)(yb + ((u64)(long long)(k + 1000) << 52))) * 0x1.0p-1000;
    const double huge = y * 2.0 * 0x1.0p1023;
    double res = k >= -1021 ? normal : tiny;
    res = k == 1024 ? huge : res;
    res = under ? 0.0 : res;
    res = over ? __builtin_inf() : res;
    return isnan ? xin : res;
}

// p*z + c with the constant c held in a scalar register pair: keeps the 21 Taylor coefficients out of the vector
// register file (hipcc otherwise parks each in a VGPR pair and copies it before every v_fmac)
__device__ __forceinline__ double fma_sconst(double p, double z, double c)
{
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(p), "v"(z), "s"(c));
    return r;
}

// cos(2 pi u) and sin(2 pi u), u in [0,1): exact quarter-turn reduction, Taylor in r
__device__ __forceinline__ void det_sincos2pi(double u, double &sn_out, double &cs_out)
{
    const double a = 4.0 * u;
    const double q = __builtin_floor(a + 0.5);
    const double r = a - q;
    const double z = r * r;
    const int qi = (int)q & 3;
    double ps = -0x1.8a404211f9547p-45;
    ps = fma_sconst(ps, z, 0x1.aaec32af93359p-38);
    ps = fma_sconst(ps, z, -0x1.6fadb9f155744p-31);
    ps = fma_sconst(ps, z, 0x1.e8f434d018d63p-25);
    ps = fma_sconst(ps, z, -0x1.e3074fde8871fp-19);
    ps = fma_sconst(ps, z, 0x1.50783487ee782p-13);
    ps = fma_sconst(ps, z, -0x1.32d2cce62bd86p-8);
    ps = fma_sconst(ps, z, 0x1.466bc6775aae2p-4);
    ps = fma_sconst(ps, z, -0x1.4abbce625be53p-1);
    ps = fma_sconst(ps, z, 0x1.921fb54442d18p+0);
    const double sn = ps * r;       // sin(pi r / 2)
    double pc = 0x1.ef6e308d6d1c4p-49;
    pc = fma_sconst(pc, z, -0x1.2a0c591af8314p-41);
    pc = fma_sconst(pc, z, 0x1.20c62c2f2d7f5p-34);
    pc = fma_sconst(pc, z, -0x1.b6e24f44b128fp-28);
    pc = fma_sconst(pc, z, 0x1.f9d38a3763cc3p-22);
    pc = fma_sconst(pc, z, -0x1.a6d1f2a204a8cp-16);
    pc = fma_sconst(pc, z, 0x1.e1f506891babbp-11);
    pc = fma_sconst(pc, z, -0x1.55d3c7e3cbffap-6);
    pc = fma_sconst(pc, z, 0x1.03c1f081b5ac4p-2);
    pc = fma_sconst(pc, z, -0x1.3bd3cc9be45dep+0);
    pc = fma_sconst(pc, z, 1.0);  // cos(pi r / 2)
    const double cv = (qi & 1) ? sn : pc;
    cs_out = (qi == 1 || qi == 2) ? -cv : cv;
    const double sv = (qi & 1) ? pc : sn;
    sn_out = (qi >= 2) ? -sv : sv;
}
__device__ __forceinline__ double det_cos2pi(double u)
{
    double s, c;
    det_sincos2pi(u, s, c);
    return c;
}
__device__ __forceinline__ double det_sin2pi(double u)
{
    double s, c;
    det_sincos2pi(u, s, c);
    return s;
}

__device__ __forceinline__ double det_sqrt(double x) { return __dsqrt_rn(x); }

// ------------------------------------------ the draws of the MH path (table driven)
// The 1 KB of tables (ptmi_tables.h, tools/make_draw_tables.py) is read from the block's LDS copy at smem[off ...] when
// the kernel has one (off >= 0: draw_table_fill), else from global memory; the values are the same.
// TM (compile time): 0 = global tables, 1 = the LDS copy, 2 = the LDS copy when off >= 0 (wave-uniform branch)
template <int TM>
__device__ __forceinline__ ptmi_dev_d2 draw_table(const double *smem, int off, u32 entry)
{
    // loads in their own address spaces: hipcc otherwise selects between the POINTERS and issues one flat load
    typedef __attribute__((address_space(3))) const ptmi_dev_d2 lds_d2;
    typedef __attribute__((address_space(1))) const ptmi_dev_d2 glb_d2;
    if (TM == 1 || (TM == 2 && off >= 0)) return *(lds_d2 *)(smem + off + 2 * entry);
    return *(glb_d2 *)(PTMI_DRAWT + 2 * entry);
}
__device__ __forceinline__ void draw_table_fill(double *smem, int off, int nthreads)
{
    if (off < 0) return;
    for (int i = (int)threadIdx.x; i < 128; i += nthreads) smem[off + i] = PTMI_DRAWT[i];
}
// ln u of the (0,1] uniform u = ((w >> 11) + 1) 2^-53 of a 64-bit word.  x = (double)n = z 2^k with z in
// [0.6953125, 1.390625) cut into 32 slices by bit pattern; r = z invc - 1 (one rounding, |r| <= 1/64),
// ln = k ln2 + logc + log1p(r), log1p by its Taylor polynomial to r^9 (next term < 6e-18 r).  The slice around z = 1 has
// invc = 1, logc = 0: a u just below 1 gives r < 0 exactly and a result that is never positive.  About 25 instructions,
// no division, no special case (det_log: about 60 with the conversion).  Oracle: orc_unit_log.
// In three pieces so that a caller can put the table reads first and the work that does not need them behind
// (DrawBatch::refill: from global memory a read takes a few hundred cycles).
struct UnitLogArg { double z; int k; u32 slice; };
__device__ __forceinline__ UnitLogArg unit_log_arg(u64 w)
{
    const u64 n = (w >> 11) + 1ull;
    const double x = __builtin_fma((double)(u32)(n >> 32), 0x1.0p32, (double)(u32)n);      // exact: n <= 2^53
    const u64 xb = (u64)__double_as_longlong(x);
    const u32 hi = (u32)(xb >> 32), tmp = hi - 0x3FE64000u;
    UnitLogArg g;
    g.k = ((int)tmp >> 20) - 53;
    g.slice = (tmp >> 15) & 31u;
    g.z = __longlong_as_double((long long)(((u64)(hi - (tmp & 0xFFF00000u)) << 32) | (u32)xb));
    return g;
}
__device__ __forceinline__ double unit_log_finish(const UnitLogArg &g, ptmi_dev_d2 e)
{
    const double r = __builtin_fma(g.z, e.x, -1.0);
    double p = 0x1.c71c71c71c71cp-4;                    // +1/9
    p = __builtin_fma(p, r, -0x1.0p-3);                 // -1/8
    p = __builtin_fma(p, r, 0x1.2492492492492p-3);      // +1/7
    p = __builtin_fma(p, r, -0x1.5555555555555p-3);     // -1/6
    p = __builtin_fma(p, r, 0x1.999999999999ap-3);      // +1/5
    p = __builtin_fma(p, r, -0x1.0p-2);                 // -1/4
    p = __builtin_fma(p, r, 0x1.5555555555555p-2);      // +1/3
    p = __builtin_fma(p, r, -0x1.0p-1);                 // -1/2
    const double l1 = __builtin_fma(r * r, p, r);
    return __builtin_fma((double)g.k, 0x1.62e42fefa39efp-1, e.y) + l1;
}
template <int TM = 0>
__device__ __forceinline__ double unit_log(u64 w, const double *smem = nullptr, int off = -1)
{
    const UnitLogArg g = unit_log_arg(w);
    return unit_log_finish(g, draw_table<TM>(smem, off, g.slice));
}
// The Box-Muller angle: 2 pi (j + 1/2 + t) / 32, j the top 5 bits of the word, t in [-1/2, 1/2) from the bits below them
__device__ __forceinline__ void unit_angle64(u64 w, u32 &j, double &t)
{
    j = (u32)(w >> 59);
    t = __longlong_as_double((long long)(((w >> 7) & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull)) - 1.5;
}
__device__ __forceinline__ void unit_angle32(u32 h, u32 &j, double &t)
{
    j = h >> 27;
    const u32 f = h & 0x07FFFFFFu;
    t = __longlong_as_double((long long)(((u64)(0x3FF00000u | (f >> 7)) << 32) | (u64)(f << 25))) - 1.5;
}
// cos and sin of that angle: the base angle's pair from the table (exactly mirrored over the octants), rotated by
// beta = 2 pi t / 32 (|beta| <= pi/32: sin to beta^9, cos to beta^8, next terms < 3e-17).  Oracle: orc_unit_sincos*.
__device__ __forceinline__ void unit_rotation(double t, double &sb, double &cb)      // sin and cos of beta: no table
{
    const double be = t * 0x1.921fb54442d18p-3, zz = be * be;
    double ps = 0x1.71de3a556c734p-19;                  // 1/9!
    ps = __builtin_fma(ps, zz, -0x1.a01a01a01a01ap-13); // -1/7!
    ps = __builtin_fma(ps, zz, 0x1.1111111111111p-7);   // 1/5!
    ps = __builtin_fma(ps, zz, -0x1.5555555555555p-3);  // -1/3!
    sb = __builtin_fma(be * zz, ps, be);
    double pc = 0x1.a01a01a01a01ap-16;                  // 1/8!
    pc = __builtin_fma(pc, zz, -0x1.6c16c16c16c17p-10); // -1/6!
    pc = __builtin_fma(pc, zz, 0x1.5555555555555p-5);   // 1/4!
    pc = __builtin_fma(pc, zz, -0x1.0p-1);              // -1/2!
    cb = __builtin_fma(zz, pc, 1.0);
}
__device__ __forceinline__ double unit_cos_finish(ptmi_dev_d2 b, double sb, double cb) { return __builtin_fma(-b.y, sb, b.x * cb); }
__device__ __forceinline__ double unit_sin_finish(ptmi_dev_d2 b, double sb, double cb) { return __builtin_fma(b.x, sb, b.y * cb); }
template <int TM = 0>
__device__ __forceinline__ void unit_sincos(u32 j, double t, double &sn, double &cs, const double *smem = nullptr, int off = -1)
{
    const ptmi_dev_d2 b = draw_table<TM>(smem, off, 32u + j);
    double sb, cb;
    unit_rotation(t, sb, cb);
    cs = unit_cos_finish(b, sb, cb);
    sn = unit_sin_finish(b, sb, cb);
}

// Box-Muller, cos branch, by the generic functions (the gradient jumps' momenta: ptmi_gj.inc.h draws its own pairs)
__device__ __forceinline__ double det_normal(u64 w0, u64 w1)
{
    return det_sqrt(-2.0 * det_log(w2uniform_open(w0))) * det_cos2pi(w2uniform(w1));
}

}  // namespace ptmi
