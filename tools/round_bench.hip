// Time of one pairing "round" (pb_round) in isolation: one workgroup, N dependent rounds; and of the single-lane Fq
// inversion inside pb_inv.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -o build/round_bench tools/round_bench.hip && ./build/round_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../snarkos_amd/csrc/pairing_wg.hpp"
using namespace mi;

template <int MODE>
#ifndef RB_THREADS
#define RB_THREADS PW_THREADS
#endif
__global__ void __launch_bounds__(RB_THREADS) rounds_kernel(int n, Fq *out) {
    extern __shared__ uint4 pw_lds[];
    LShared &S = *(LShared *)pw_lds;
    const int t = threadIdx.x;
    if (t < 12) {
        Fq v = Fq::one();
        v.l[0] += 7 * t + 3;
        pb_put(&S.f, t, v);
        pb_put(&S.t, t, v);
    }
    if (t == 12) {
        pb_put_scale(&S.f, Fq::one());
        pb_put_scale(&S.t, Fq::one());
    }
    if (t >= 64 && t < 64 + 25) {  // any constants below q do for timing
        Fq v = Fq::one();
        v.l[2] ^= t;
        lds_store(&S.dtab[t - 64], v);
    }
    __syncthreads();
    LPB *cur = &S.f, *alt = &S.g;
    if (MODE == 2) {
        Fq x = Fq::one();
        x.l[0] += t;
        for (int i = 0; i < n; i++) {
            if (t == 0) x = invq(x);
            __syncthreads();
        }
        if (t == 0) out[0] = x;
        return;
    }
    if (MODE == 3) {  // stage profile of the ping-pong round (fold constants in registers)
        __shared__ uint64_t prof[8];
        if (t < 8) prof[t] = 0;
        __syncthreads();
        const PwFold fc = pw_fold_consts(S);
        for (int i = 0; i < n; i++) {
            pb_round_impl<true, true>(S, alt, cur->c, cur, fc.d0, fc.d1, fc.d2, fc.d3, prof);
            LPB *sw = cur; cur = alt; alt = sw;
        }
        if (t < 8) ((uint64_t *)out)[80 + t] = prof[t];  // behind the 12 result coefficients (576 B)
        if (t < 12) out[t] = lds_load(&cur->c[t]);
        return;
    }
    if (MODE == 5 || MODE == 6) {  // the cyclotomic squaring round (timing only: the values are no subgroup members)
        __shared__ uint64_t prof[8];
        if (t < 8) prof[t] = 0;
        __syncthreads();
        const PwFold fc = pw_fold_consts(S);
        const uint32_t entry = pw_csqr_entry();
        for (int i = 0; i < n; i++) {
            pb_csqr_impl<MODE == 6>(S, alt, cur, entry, fc.d0, fc.d1, fc.d2, fc.d3, prof);
            LPB *sw = cur; cur = alt; alt = sw;
        }
        if (MODE == 6 && t < 8) ((uint64_t *)out)[80 + t] = prof[t];
        if (t < 12) out[t] = lds_load(&cur->c[t]);
        return;
    }
    if (MODE == 4) {  // ping-pong rounds with the fold constants kept in registers
        const PwFold fc = pw_fold_consts(S);
        for (int i = 0; i < n; i++) {
            pb_round_impl<true>(S, alt, cur->c, cur, fc.d0, fc.d1, fc.d2, fc.d3);
            LPB *sw = cur; cur = alt; alt = sw;
        }
        if (t < 12) out[t] = lds_load(&cur->c[t]);
        return;
    }
    for (int i = 0; i < n; i++) {
        if (MODE == 0) pb_mul(S, &S.f, &S.f, &S.f);
        else {
            pb_mul_na(S, alt, cur, cur);
            LPB *sw = cur; cur = alt; alt = sw;
        }
    }
    if (t < 12) out[t] = lds_load(&cur->c[t]);
}

int main() {
    Fq *d;
    hipMalloc(&d, sizeof(Fq) * 12 + 1024);
    hipMemset(d, 0, sizeof(Fq) * 12 + 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const char *names[7] = {"dense round, in place (2 barriers)", "dense round, ping-pong (1 barrier)", "Fq inversion on one lane",
                            "ping-pong round with stage stamps", "ping-pong round, fold constants in registers",
                            "cyclotomic squaring round", "cyclotomic squaring round with stage stamps"};
    hipFuncSetAttribute((const void *)rounds_kernel<5>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PW_LDS_BYTES);
    hipFuncSetAttribute((const void *)rounds_kernel<6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PW_LDS_BYTES);
    hipFuncSetAttribute((const void *)rounds_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PW_LDS_BYTES);
    hipFuncSetAttribute((const void *)rounds_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PW_LDS_BYTES);
    hipFuncSetAttribute((const void *)rounds_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PW_LDS_BYTES);
    hipFuncSetAttribute((const void *)rounds_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PW_LDS_BYTES);
    hipFuncSetAttribute((const void *)rounds_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PW_LDS_BYTES);
    for (int mode = 0; mode < 7; mode++) {
        const int N = mode == 2 ? 50 : 2000;
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0, 0);
            if (mode == 0) hipLaunchKernelGGL(rounds_kernel<0>, dim3(1), dim3(RB_THREADS), PW_LDS_BYTES, 0, N, d);
            else if (mode == 1) hipLaunchKernelGGL(rounds_kernel<1>, dim3(1), dim3(RB_THREADS), PW_LDS_BYTES, 0, N, d);
            else if (mode == 2) hipLaunchKernelGGL(rounds_kernel<2>, dim3(1), dim3(RB_THREADS), PW_LDS_BYTES, 0, N, d);
            else if (mode == 3) hipLaunchKernelGGL(rounds_kernel<3>, dim3(1), dim3(RB_THREADS), PW_LDS_BYTES, 0, N, d);
            else if (mode == 4) hipLaunchKernelGGL(rounds_kernel<4>, dim3(1), dim3(RB_THREADS), PW_LDS_BYTES, 0, N, d);
            else if (mode == 5) hipLaunchKernelGGL(rounds_kernel<5>, dim3(1), dim3(RB_THREADS), PW_LDS_BYTES, 0, N, d);
            else hipLaunchKernelGGL(rounds_kernel<6>, dim3(1), dim3(RB_THREADS), PW_LDS_BYTES, 0, N, d);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("%s: %.3f us each\n", names[mode], ms * 1e3 / N);
            if (rep && (mode == 3 || mode == 6)) {
                uint64_t h[8];
                hipMemcpy(h, (char *)d + 80 * 8, sizeof h, hipMemcpyDeviceToHost);
                const char *st[7] = {"loads", "product", "quad reduction", "fold", "row reduction", "Barrett + twin", "stores + barrier"};
                for (int k = 0; k < 7; k++) printf("    %-18s %7.1f ticks per round\n", st[k], (double)h[k] / N);
            }
        }
    }
    return 0;
}
