// Time of one pairing "round" (pb_round) in isolation: one workgroup, N dependent rounds.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -o build/round_bench tools/round_bench.hip && ./build/round_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../snarkos_amd/csrc/pairing_wg.hpp"
using namespace mi;

template <int MODE>
__global__ void __launch_bounds__(PW_THREADS) rounds_kernel(int n, Fq *out) {
    __shared__ PwShared S;
    const int t = threadIdx.x;
    if (t < 12) {
        Fq v = Fq::one();
        v.l[0] += 7 * t + 3;
        lds_store(&S.f.c[t], v);
        lds_store(&S.t.c[t], v);
    }
    if (t < 8) {
        Fq v = Fq::one();
        v.l[1] += t;
        lds_store(&S.lines2[0][t >> 2][t & 3], v);
    }
    if (t == 12) { lds_store(&S.py[0], Fq::one()); lds_store(&S.py[1], Fq::one()); }
    __syncthreads();
    PB *cur = &S.f, *alt = &S.g;
    for (int i = 0; i < n; i++) {
        if (MODE == 0) pb_mul(S, &S.f, &S.f, &S.f);
        else if (MODE == 1) pb_round<true>(S, &S.f, &S.f, &S.f, &S.t, &S.py[0], S.lines2[0][0], &S.py[1], S.lines2[0][1]);
        else {
            if (MODE == 2) pb_mul_na(S, alt, cur, cur);
            else pb_round<true, true>(S, alt, cur, cur, &S.t, &S.py[0], S.lines2[0][0], &S.py[1], S.lines2[0][1]);
            PB *sw = cur; cur = alt; alt = sw;
        }
    }
    if (t < 12) out[t] = lds_load(&cur->c[t]);
}

int main() {
    Fq *d;
    hipMalloc(&d, sizeof(Fq) * 12);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int N = 2000;
    const char *names[4] = {"dense round, in place (2 barriers)", "round with line product, in place", "dense round, ping-pong (1 barrier)",
                            "round with line product, ping-pong"};
    for (int mode = 0; mode < 4; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0, 0);
            if (mode == 0) hipLaunchKernelGGL(rounds_kernel<0>, dim3(1), dim3(PW_THREADS), 0, 0, N, d);
            else if (mode == 1) hipLaunchKernelGGL(rounds_kernel<1>, dim3(1), dim3(PW_THREADS), 0, 0, N, d);
            else if (mode == 2) hipLaunchKernelGGL(rounds_kernel<2>, dim3(1), dim3(PW_THREADS), 0, 0, N, d);
            else hipLaunchKernelGGL(rounds_kernel<3>, dim3(1), dim3(PW_THREADS), 0, 0, N, d);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("%s: %.3f us per round\n", names[mode], ms * 1e3 / N);
        }
    }
    return 0;
}
