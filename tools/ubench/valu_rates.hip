// Developer tool: issue cost of the instructions the conditioning-set kernel is made of, on one wavefront (gfx950).
// Each kernel runs REP x 64 independent copies of one instruction on 8 accumulators; the table printed is ns per
// wave-instruction and the ratio to v_fma_f64.     hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>
#define REP 2000
#define K8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define K64(X) K8(X) K8(X) K8(X) K8(X) K8(X) K8(X) K8(X) K8(X)

#define DEFK(NAME, ASMSTR)                                                                        \
    __global__ void __launch_bounds__(64) NAME(double *out, double x, double y)                   \
    {                                                                                             \
        double a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3, a4 = x + 4, a5 = x + 5, a6 = x + 6, a7 = x + 7; \
        double b = y + threadIdx.x, c = y * 0.5;                                                  \
        int n0 = threadIdx.x, n1 = n0 + 1, n2 = n0 + 2, n3 = n0 + 3, n4 = n0 + 4, n5 = n0 + 5, n6 = n0 + 6, n7 = n0 + 7; \
        int nb = (int)x + threadIdx.x, nc = (int)y;                                               \
        for (int r = 0; r < REP; ++r) {                                                           \
            K64(ASMSTR)                                                                           \
        }                                                                                         \
        out[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (n0 + n1 + n2 + n3 + n4 + n5 + n6 + n7 + nb); \
    }
#define OP_FMA(i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a##i) : "v"(b), "v"(c));
#define OP_FMAC_DPP(i) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a##i) : "v"(b), "v"(c));
#define OP_MOV64_DPP(i) asm volatile("v_mov_b64_dpp %0, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a##i));
#define OP_MOV32_DPP(i) asm volatile("v_mov_b32_dpp %0, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(n##i));
#define OP_MUL(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a##i) : "v"(c));
#define OP_ADD(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a##i) : "v"(c));
#define OP_RCP(i) asm volatile("v_rcp_f64 %0, %0" : "+v"(a##i));
#define OP_RSQ(i) asm volatile("v_rsq_f64 %0, %0" : "+v"(a##i));
#define OP_LDEXP(i) asm volatile("v_ldexp_f64 %0, %0, 1" : "+v"(a##i));
#define OP_RNDNE(i) asm volatile("v_rndne_f64 %0, %0" : "+v"(a##i));
#define OP_CVT(i) asm volatile("v_cvt_i32_f64 %0, %1" : "+v"(n##i) : "v"(a##i));
#define OP_MAX(i) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a##i) : "v"(c));
#define OP_CND(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(n##i) : "v"(nc));
#define OP_MOV32(i) asm volatile("v_mov_b32 %0, %0" : "+v"(n##i));
#define OP_ADD32(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(n##i) : "v"(nc));
#define OP_MAD24(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(n##i) : "v"(nc));
#define OP_FMA32(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(n##i) : "v"(nb), "v"(nc));
#define OP_PKFMA32(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a##i) : "v"(b), "v"(c));
#define OP_CMP(i) asm volatile("v_cmp_gt_f64 vcc, %0, %1" : : "v"(b), "v"(c) : "vcc");
#define OP_PERM(i) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(n##i), "+v"(nb));
DEFK(k_fma, OP_FMA) DEFK(k_fmac_dpp, OP_FMAC_DPP) DEFK(k_mov64_dpp, OP_MOV64_DPP) DEFK(k_mov32_dpp, OP_MOV32_DPP)
DEFK(k_mul, OP_MUL) DEFK(k_add, OP_ADD) DEFK(k_rcp, OP_RCP) DEFK(k_rsq, OP_RSQ) DEFK(k_ldexp, OP_LDEXP)
DEFK(k_rndne, OP_RNDNE) DEFK(k_cvt, OP_CVT) DEFK(k_max, OP_MAX) DEFK(k_cnd, OP_CND) DEFK(k_mov32, OP_MOV32)
DEFK(k_add32, OP_ADD32) DEFK(k_mad24, OP_MAD24) DEFK(k_fma32, OP_FMA32) DEFK(k_pkfma32, OP_PKFMA32) DEFK(k_cmp, OP_CMP)
DEFK(k_perm, OP_PERM)

int main()
{
    double *d;
    hipMalloc(&d, 64 * 8);
    struct E { const char *n; void (*f)(double *, double, double); };
    E es[] = {{"v_fma_f64", k_fma}, {"v_fmac_f64_dpp row_newbcast", k_fmac_dpp}, {"v_mov_b64_dpp", k_mov64_dpp},
              {"v_mov_b32_dpp", k_mov32_dpp}, {"v_mul_f64", k_mul}, {"v_add_f64", k_add}, {"v_rcp_f64", k_rcp},
              {"v_rsq_f64", k_rsq}, {"v_ldexp_f64", k_ldexp}, {"v_rndne_f64", k_rndne}, {"v_cvt_i32_f64", k_cvt},
              {"v_max_f64", k_max}, {"v_cndmask_b32", k_cnd}, {"v_mov_b32", k_mov32}, {"v_add_u32", k_add32},
              {"v_mad_u32_u24", k_mad24}, {"v_fma_f32", k_fma32}, {"v_pk_fma_f32", k_pkfma32}, {"v_cmp_gt_f64", k_cmp},
              {"v_permlane16_swap_b32", k_perm}};
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    double base = 0;
    for (auto &e : es) {
                for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(e.f, dim3(1), dim3(64), 0, 0, d, 1.0, 2.0);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(e.f, dim3(1), dim3(64), 0, 0, d, 1.0, 2.0);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double ns = ms * 1e6 / 5 / (double(REP) * 64);
        if (base == 0) base = ns;
        printf("%-32s %7.3f ns/inst   x%.2f of v_fma_f64\n", e.n, ns, ns / base);
    }
    return 0;
}
