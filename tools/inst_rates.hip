// Issue cost of the instructions the step kernels are made of, alone and beside f64 matrix instructions (developer tool).
// Every test is one asm body on fixed registers, repeated in a loop and timed with the shader clock by every wave;
// blocks of 256 threads = one wave per SIMD, blocks of 512 = two.  Prints cycles per body per SIMD.
// hipcc --offload-arch=gfx950 -O3 tools/inst_rates.hip -o /tmp/inst_rates && /tmp/inst_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define R2(x) x x
#define R4(x) R2(x) R2(x)
#define R8(x) R4(x) R4(x)
#define R16(x) R8(x) R8(x)
#define CLOB "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", \
             "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", \
             "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", \
             "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", \
             "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", \
             "a62", "a63", "vcc", "s20", "s21"

// eight independent destinations v[20:35] (pairs), sources v[40:47]
#define F64_8(op) \
    op " v[20:21], v[40:41], v[42:43]\n" op " v[22:23], v[40:41], v[42:43]\n" op " v[24:25], v[40:41], v[42:43]\n" op " v[26:27], v[40:41], v[42:43]\n" \
    op " v[28:29], v[40:41], v[42:43]\n" op " v[30:31], v[40:41], v[42:43]\n" op " v[32:33], v[40:41], v[42:43]\n" op " v[34:35], v[40:41], v[42:43]\n"
#define F64_8_3(op) \
    op " v[20:21], v[40:41], v[42:43], v[20:21]\n" op " v[22:23], v[40:41], v[42:43], v[22:23]\n" op " v[24:25], v[40:41], v[42:43], v[24:25]\n" \
    op " v[26:27], v[40:41], v[42:43], v[26:27]\n" op " v[28:29], v[40:41], v[42:43], v[28:29]\n" op " v[30:31], v[40:41], v[42:43], v[30:31]\n" \
    op " v[32:33], v[40:41], v[42:43], v[32:33]\n" op " v[34:35], v[40:41], v[42:43], v[34:35]\n"
#define F64_8_1(op) \
    op " v[20:21], v[40:41]\n" op " v[22:23], v[40:41]\n" op " v[24:25], v[40:41]\n" op " v[26:27], v[40:41]\n" \
    op " v[28:29], v[40:41]\n" op " v[30:31], v[40:41]\n" op " v[32:33], v[40:41]\n" op " v[34:35], v[40:41]\n"
#define U32_8(op) \
    op " v20, v40, v41\n" op " v21, v40, v41\n" op " v22, v40, v41\n" op " v23, v40, v41\n" op " v24, v40, v41\n" op " v25, v40, v41\n" op " v26, v40, v41\n" \
    op " v27, v40, v41\n"
#define MAD64_8 \
    "v_mad_u64_u32 v[20:21], vcc, v40, v41, 0\n" "v_mad_u64_u32 v[22:23], vcc, v40, v41, 0\n" "v_mad_u64_u32 v[24:25], vcc, v40, v41, 0\n" \
    "v_mad_u64_u32 v[26:27], vcc, v40, v41, 0\n" "v_mad_u64_u32 v[28:29], vcc, v40, v41, 0\n" "v_mad_u64_u32 v[30:31], vcc, v40, v41, 0\n" \
    "v_mad_u64_u32 v[32:33], vcc, v40, v41, 0\n" "v_mad_u64_u32 v[34:35], vcc, v40, v41, 0\n"
#define BITOP_8 \
    "v_bitop3_b32 v20, v40, v41, v42 bitop3:0x96\n" "v_bitop3_b32 v21, v40, v41, v42 bitop3:0x96\n" "v_bitop3_b32 v22, v40, v41, v42 bitop3:0x96\n" \
    "v_bitop3_b32 v23, v40, v41, v42 bitop3:0x96\n" "v_bitop3_b32 v24, v40, v41, v42 bitop3:0x96\n" "v_bitop3_b32 v25, v40, v41, v42 bitop3:0x96\n" \
    "v_bitop3_b32 v26, v40, v41, v42 bitop3:0x96\n" "v_bitop3_b32 v27, v40, v41, v42 bitop3:0x96\n"
#define CVT_8 \
    "v_cvt_f64_u32 v[20:21], v40\n" "v_cvt_f64_u32 v[22:23], v40\n" "v_cvt_f64_u32 v[24:25], v40\n" "v_cvt_f64_u32 v[26:27], v40\n" \
    "v_cvt_f64_u32 v[28:29], v40\n" "v_cvt_f64_u32 v[30:31], v40\n" "v_cvt_f64_u32 v[32:33], v40\n" "v_cvt_f64_u32 v[34:35], v40\n"
#define LDEXP_8 \
    "v_ldexp_f64 v[20:21], v[40:41], 3\n" "v_ldexp_f64 v[22:23], v[40:41], 3\n" "v_ldexp_f64 v[24:25], v[40:41], 3\n" "v_ldexp_f64 v[26:27], v[40:41], 3\n" \
    "v_ldexp_f64 v[28:29], v[40:41], 3\n" "v_ldexp_f64 v[30:31], v[40:41], 3\n" "v_ldexp_f64 v[32:33], v[40:41], 3\n" "v_ldexp_f64 v[34:35], v[40:41], 3\n"
#define ACCRW_8 \
    "v_accvgpr_write_b32 a56, v40\n" "v_accvgpr_read_b32 v20, a57\n" "v_accvgpr_write_b32 a58, v40\n" "v_accvgpr_read_b32 v21, a59\n" \
    "v_accvgpr_write_b32 a60, v40\n" "v_accvgpr_read_b32 v22, a61\n" "v_accvgpr_write_b32 a62, v40\n" "v_accvgpr_read_b32 v23, a63\n"
// seven matrix instructions on seven accumulators (the AM product's k-step), with a filler behind each
#define MM(acc, fill) "v_mfma_f64_16x16x4_f64 " acc ", v[44:45], v[46:47], " acc "\n" fill
#define MM7(fill) MM("a[0:7]", fill) MM("a[8:15]", fill) MM("a[16:23]", fill) MM("a[24:31]", fill) MM("a[32:39]", fill) MM("a[40:47]", fill) MM("a[48:55]", fill)
#define XOR1 "v_xor_b32 v20, v40, v41\n"
#define MULHI1 "v_mul_hi_u32 v21, v40, v41\n"
#define MAD1 "v_mad_u64_u32 v[22:23], vcc, v40, v41, 0\n"
#define FMA1 "v_fma_f64 v[24:25], v[40:41], v[42:43], v[24:25]\n"
#define FMA1B "v_fma_f64 v[26:27], v[40:41], v[42:43], v[26:27]\n"
#define ROUND MAD1 "v_mad_u64_u32 v[28:29], vcc, v40, v42, 0\n" "v_bitop3_b32 v30, v40, v41, v42 bitop3:0x96\n" "v_bitop3_b32 v31, v40, v41, v42 bitop3:0x96\n"
#define ROUNDM "v_mul_hi_u32 v21, v40, v41\n" "v_mul_lo_u32 v22, v40, v41\n" "v_mul_hi_u32 v23, v40, v42\n" "v_mul_lo_u32 v28, v40, v42\n" \
               "v_bitop3_b32 v30, v40, v41, v42 bitop3:0x96\n" "v_bitop3_b32 v31, v40, v41, v42 bitop3:0x96\n"
#define DSR "ds_read_b64 v[48:49], v50\n"

struct Test { const char *name; int ninst; };
#define TESTS(X) \
    X(0, "v_fma_f64 x8", 8, F64_8_3("v_fma_f64")) \
    X(1, "v_mul_f64 x8", 8, F64_8("v_mul_f64")) \
    X(2, "v_add_f64 x8", 8, F64_8("v_add_f64")) \
    X(3, "v_mad_u64_u32 x8", 8, MAD64_8) \
    X(4, "v_mul_hi_u32 x8", 8, U32_8("v_mul_hi_u32")) \
    X(5, "v_mul_lo_u32 x8", 8, U32_8("v_mul_lo_u32")) \
    X(6, "v_bitop3_b32 x8", 8, BITOP_8) \
    X(7, "v_xor_b32 x8", 8, U32_8("v_xor_b32")) \
    X(8, "v_rcp_f64 x8", 8, F64_8_1("v_rcp_f64")) \
    X(9, "v_rsq_f64 x8", 8, F64_8_1("v_rsq_f64")) \
    X(10, "v_cvt_f64_u32 x8", 8, CVT_8) \
    X(11, "v_ldexp_f64 x8", 8, LDEXP_8) \
    X(12, "v_floor_f64 x8", 8, F64_8_1("v_floor_f64")) \
    X(13, "accvgpr write/read x8", 8, ACCRW_8) \
    X(14, "v_div_scale-free: v_div_fixup_f64 x8", 8, F64_8_3("v_div_fixup_f64")) \
    X(15, "mfma x7", 7, MM7("")) \
    X(16, "mfma x7 + 4 xor each", 7, MM7(R4(XOR1))) \
    X(17, "mfma x7 + 8 xor each", 7, MM7(R8(XOR1))) \
    X(18, "mfma x7 + 12 xor each", 7, MM7(R8(XOR1) R4(XOR1))) \
    X(19, "mfma x7 + 16 xor each", 7, MM7(R16(XOR1))) \
    X(20, "mfma x7 + 1 mul_hi each", 7, MM7(MULHI1)) \
    X(21, "mfma x7 + 2 mul_hi each", 7, MM7(R2(MULHI1))) \
    X(22, "mfma x7 + 4 mul_hi each", 7, MM7(R4(MULHI1))) \
    X(23, "mfma x7 + philox round (2 mad64 + 2 bitop3) each", 7, MM7(ROUND)) \
    X(24, "mfma x7 + philox round (4 mul + 2 bitop3) each", 7, MM7(ROUNDM)) \
    X(25, "mfma x7 + 2 rounds each", 7, MM7(ROUND ROUND)) \
    X(26, "mfma x7 + 1 fma_f64 each", 7, MM7(FMA1)) \
    X(27, "mfma x7 + 2 fma_f64 each", 7, MM7(FMA1 FMA1B)) \
    X(28, "mfma x7 + 4 fma_f64 each", 7, MM7(FMA1 FMA1B FMA1 FMA1B)) \
    X(29, "mfma x7 + 8 fma_f64 each", 7, MM7(R4(FMA1 FMA1B))) \
    X(30, "mfma x7 + ds_read_b64 each", 7, MM7(DSR)) \
    X(31, "philox round x7 alone (mad64)", 7, R4(ROUND) R2(ROUND) ROUND) \
    X(32, "philox round x7 alone (4 mul)", 7, R4(ROUNDM) R2(ROUNDM) ROUNDM) \
    X(33, "v_cndmask_b32 x8", 8, U32_8("v_cndmask_b32")) \
    X(34, "v_fma_f64 dependent chain x8", 8, R8("v_fma_f64 v[20:21], v[20:21], v[42:43], v[40:41]\n")) \
    X(35, "v_fma_f64 two chains x8", 8, R4("v_fma_f64 v[20:21], v[20:21], v[42:43], v[40:41]\nv_fma_f64 v[22:23], v[22:23], v[42:43], v[40:41]\n")) \
    X(36, "v_mul_f64 dependent chain x8", 8, R8("v_mul_f64 v[20:21], v[20:21], v[42:43]\n")) \
    X(37, "v_mad_u64_u32 dependent x8", 8, R8("v_mad_u64_u32 v[20:21], vcc, v20, v41, 0\n")) \
    X(38, "v_cndmask_b32_e64 sgpr mask x8", 8, R8("v_cndmask_b32_e64 v20, v40, v41, s[20:21]\n")) \
    X(39, "v_cndmask_b32 vcc, after v_cmp x (cmp+cnd)x4", 8, R4("v_cmp_gt_f64 vcc, v[40:41], v[42:43]\nv_cndmask_b32 v20, v40, v41, vcc\n")) \
    X(40, "v_cmp_gt_f64 vcc x8", 8, R8("v_cmp_gt_f64 vcc, v[40:41], v[42:43]\n")) \
    X(41, "v_cmp_gt_u32 sgpr x8", 8, R8("v_cmp_gt_u32 s[20:21], v40, v41\n")) \
    X(42, "v_readlane_b32 x8", 8, R8("v_readlane_b32 s20, v40, 3\n")) \
    X(43, "v_div_scale_f64 x8", 8, R8("v_div_scale_f64 v[20:21], vcc, v[40:41], v[42:43], v[40:41]\n")) \
    X(44, "v_div_fmas_f64 x8", 8, R8("v_div_fmas_f64 v[20:21], v[40:41], v[42:43], v[40:41]\n")) \
    X(45, "s_and_saveexec + s_mov exec x4", 8, R4("s_and_saveexec_b64 s[20:21], vcc\ns_mov_b64 exec, s[20:21]\n")) \
    X(46, "v_cvt_i32_f64 x8", 8, R8("v_cvt_i32_f64 v20, v[40:41]\n")) \
    X(47, "v_mov_b32 dpp quad_perm x8", 8, R8("v_mov_b32_dpp v20, v40 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n")) \
    X(48, "v_fma_f64 sgpr const x8", 8, R8("v_fma_f64 v[20:21], v[40:41], v[42:43], s[20:21]\n")) \
    X(49, "v_pk_fma_f32 x8", 8, R8("v_pk_fma_f32 v[20:21], v[40:41], v[42:43], v[40:41]\n")) \
    X(50, "v_fma_f32 x8", 8, R8("v_fma_f32 v20, v40, v42, v41\n")) \
    X(51, "v_log_f32 x8", 8, R8("v_log_f32 v20, v40\n")) \
    X(52, "v_lshrrev_b64 x8", 8, R8("v_lshrrev_b64 v[20:21], 11, v[40:41]\n")) \
    X(53, "v_cndmask_b32 vcc x8 distinct regs no dep", 8, "v_cndmask_b32 v20, v40, v41, vcc\nv_cndmask_b32 v21, v42, v43, vcc\nv_cndmask_b32 v22, v44, v45, vcc\nv_cndmask_b32 v23, v46, v47, vcc\nv_cndmask_b32 v24, v40, v41, vcc\nv_cndmask_b32 v25, v42, v43, vcc\nv_cndmask_b32 v26, v44, v45, vcc\nv_cndmask_b32 v27, v46, v47, vcc\n")

template <int T>
__global__ void k(unsigned long long *out, int iters, double a, double b)
{
    __shared__ double lds[512];
    lds[threadIdx.x & 511] = a;
    __syncthreads();
    asm volatile("v_mov_b32 v40, %0\nv_mov_b32 v41, %1\nv_mov_b32 v42, %2\nv_mov_b32 v43, %3\nv_mov_b32 v44, %0\nv_mov_b32 v45, %1\nv_mov_b32 v46, %2\nv_mov_b32 v47, %3\n"
                 "v_mov_b32 v50, 0\ns_mov_b64 s[20:21], exec\ns_mov_b64 vcc, exec\n"
                 :: "v"((unsigned)__double_as_longlong(a)), "v"((unsigned)(__double_as_longlong(a) >> 32)), "v"((unsigned)__double_as_longlong(b)),
                    "v"((unsigned)(__double_as_longlong(b) >> 32)) : CLOB);
    const unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#define BODY(id, name, n, body) if constexpr (T == id) asm volatile(R4(body) "s_waitcnt lgkmcnt(0)\n" ::: CLOB, "memory");
        TESTS(BODY)
    }
    asm volatile("s_nop 15\ns_nop 15" ::: CLOB);
    const unsigned long long t1 = clock64();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int T>
void run(const char *name, int ninst, int threads)
{
    const int blocks = 256, iters = 2000;
    unsigned long long *out;
    (void)hipMalloc(&out, sizeof(unsigned long long) * blocks * 8);
    k<T><<<blocks, threads>>>(out, 10, 1.0, 0.999);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    k<T><<<blocks, threads>>>(out, iters, 1.0, 0.999);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * threads / 64);
    (void)hipMemcpy(h.data(), out, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += (double)v;
    s /= h.size();
    const double per_body = s / (iters * 4.0);                     // shader cycles per body, one wave
    const int wps = threads / 256;
    printf("%-52s %d wave/SIMD: %8.1f cyc/body/wave = %6.2f cyc per counted instr per SIMD   (wall %.3f ms -> clock %.2f GHz)\n", name, wps, per_body,
           per_body / ninst / wps, ms, s / (ms * 1e-3) / 1e9);
    (void)hipFree(out);
}
int main(int argc, char **argv)
{
#define RUN(id, name, n, body) run<id>(name, n, 256); run<id>(name, n, 512);
    TESTS(RUN)
    return 0;
}
