// Integer-ALU micro-benchmarks for gfx950: issue rates that bound the 377-bit Montgomery product.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -o build/ubench tools/ubench.hip && ./build/ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <type_traits>
#include "../snarkos_amd/csrc/field.hpp"
#include "../snarkos_amd/csrc/field28.hpp"
using namespace mi;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ void __launch_bounds__(256) alu_kernel(uint32_t *out, int iters) {
    uint32_t a = threadIdx.x * 2654435761u + 1, b = blockIdx.x * 40503u + 7;
    uint64_t acc0 = a, acc1 = b, acc2 = a ^ b, acc3 = a + b;
    uint32_t x0 = a, x1 = b, x2 = a ^ b, x3 = a + b;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            if (MODE == 0) {  // 4 independent v_mad_u64_u32 chains
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(b) : "vcc");
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc1) : "v"(a), "v"(b) : "vcc");
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc2) : "v"(a), "v"(b) : "vcc");
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc3) : "v"(a), "v"(b) : "vcc");
            } else if (MODE == 1) {  // v_add_u32
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(x0) : "v"(a));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(x1) : "v"(a));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(x2) : "v"(a));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(x3) : "v"(a));
            } else if (MODE == 2) {  // v_lshl_add_u64 (64-bit add)
                asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc0) : "v"(acc3));
                asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc1) : "v"(acc3));
                asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc2) : "v"(acc3));
                asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc3) : "v"(acc0));
            } else if (MODE == 3) {  // v_mul_lo_u32
                asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x0) : "v"(a));
                asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x1) : "v"(a));
                asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x2) : "v"(a));
                asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x3) : "v"(a));
            } else if (MODE == 4) {  // v_addc_co_u32 chain
                asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(x0) : "v"(a) : "vcc");
                asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(x1) : "v"(a) : "vcc");
                asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(x2) : "v"(a) : "vcc");
                asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(x3) : "v"(a) : "vcc");
            } else if (MODE == 5) {  // v_mad_u32_u24 (24-bit multiply-add, full rate?)
                asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x0) : "v"(a), "v"(b));
                asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x1) : "v"(a), "v"(b));
                asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x2) : "v"(a), "v"(b));
                asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x3) : "v"(a), "v"(b));
            } else if (MODE == 6) {  // v_fma_f64
                double d0 = __longlong_as_double(acc0), d1 = __longlong_as_double(acc1), d2 = __longlong_as_double(acc2), d3 = __longlong_as_double(acc3);
                asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d0) : "v"(d3));
                asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d1) : "v"(d3));
                asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d2) : "v"(d3));
                asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d2) : "v"(d3));
                acc0 = __double_as_longlong(d0); acc1 = __double_as_longlong(d1); acc2 = __double_as_longlong(d2);
            } else if (MODE == 7) {  // v_mul_hi_u32
                asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x0) : "v"(a));
                asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x1) : "v"(a));
                asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x2) : "v"(a));
                asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x3) : "v"(a));
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(acc0 ^ acc1 ^ acc2 ^ acc3) ^ x0 ^ x1 ^ x2 ^ x3;
}

template <class F, int V>
__device__ __forceinline__ F vmul(const F &a, const F &b) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (V == 0) return mul32(a, b);
    if (V == 1) return mul_comba_gen<typename F::Params, 1>(a, b);
    return mul_comba_gen<typename F::Params, 2>(a, b);
#else
    return mul32(a, b);
#endif
}
template <class F, int V>
__global__ void __launch_bounds__(256) fmul_kernel(F *out, int iters) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    F a = F::one(), b = F::r2();
    a.l[0] += (uint32_t)i;
    for (int k = 0; k < iters; k++) {
        a = vmul<F, V>(a, b);
        b = vmul<F, V>(b, a);
    }
    out[i] = add(a, b);
}

// the reduced-radix product (field28.hpp): same loop shape as fmul_kernel
template <int V>
__global__ void __launch_bounds__(256) fmul28_kernel(Fq *out, int iters) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    Fq a0 = Fq::one(), b0 = Fq::r2();
    a0.l[0] += (uint32_t)i;
    Fq28 a = load28(a0), b = load28(b0);
    for (int k = 0; k < iters; k++) {
        if (V == 0) { a = mul28(a, b); b = mul28(b, a); }
        else { a = sqr28(a); b = sqr28(sub28(b, a)); }
    }
    out[i] = store28(sub28(a, b));
}

// the same loop with only the first `active` lanes of every wave enabled: does a partly masked wave issue faster?
__global__ void __launch_bounds__(256) fmul28_masked_kernel(Fq *out, int iters, int active) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    Fq a0 = Fq::one(), b0 = Fq::r2();
    a0.l[0] += (uint32_t)i;
    Fq28 a = load28(a0), b = load28(b0);
    if ((int)(threadIdx.x & 63) < active)
        for (int k = 0; k < iters; k++) { a = mul28(a, b); b = mul28(b, a); }
    out[i] = store28(sub28(a, b));
}

// arbitrary lane mask
__global__ void __launch_bounds__(256) fmul28_lanemask_kernel(Fq *out, int iters, unsigned long long mask) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    Fq a0 = Fq::one(), b0 = Fq::r2();
    a0.l[0] += (uint32_t)i;
    Fq28 a = load28(a0), b = load28(b0);
    if ((mask >> (threadIdx.x & 63)) & 1ull)
        for (int k = 0; k < iters; k++) { a = mul28(a, b); b = mul28(b, a); }
    out[i] = store28(sub28(a, b));
}

// mixed launch: even blocks run with all lanes, odd blocks with `active` lanes; per-block duration in ticks of the
// constant 100 MHz counter -- separates an issue-rate effect of the mask from a clock (power-state) effect
__global__ void __launch_bounds__(256) fmul28_mixed_kernel(Fq *out, int iters, int active, unsigned long long *ticks) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    Fq a0 = Fq::one(), b0 = Fq::r2();
    a0.l[0] += (uint32_t)i;
    Fq28 a = load28(a0), b = load28(b0);
    const int lanes = (blockIdx.x & 1) ? active : 64;
    const unsigned long long t0 = wall_clock64();
    if ((int)(threadIdx.x & 63) < lanes)
        for (int k = 0; k < iters; k++) { a = mul28(a, b); b = mul28(b, a); }
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
    out[i] = store28(sub28(a, b));
}

template <class K, class... A>
static double time_kernel(K k, dim3 g, dim3 b, A... args) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, g, b, 0, 0, args...);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k, g, b, 0, 0, args...);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3;
}

int main() {
    uint32_t *d;
    CHECK(hipMalloc(&d, 256u * 4096 * 48));
    const char *names[] = {"v_mad_u64_u32", "v_add_u32", "v_lshl_add_u64", "v_mul_lo_u32", "v_addc_co_u32", "v_mad_u32_u24", "v_fma_f64", "v_mul_hi_u32"};
    const int iters = 2000;
    for (int waves_per_simd = 1; waves_per_simd <= 4; waves_per_simd *= 2) {
        int blocks = 256 * waves_per_simd;  // 256 CUs x (256-thread blocks = 1 wave per SIMD each)
        double t[8];
        t[0] = time_kernel(alu_kernel<0>, dim3(blocks), dim3(256), d, iters);
        t[1] = time_kernel(alu_kernel<1>, dim3(blocks), dim3(256), d, iters);
        t[2] = time_kernel(alu_kernel<2>, dim3(blocks), dim3(256), d, iters);
        t[3] = time_kernel(alu_kernel<3>, dim3(blocks), dim3(256), d, iters);
        t[4] = time_kernel(alu_kernel<4>, dim3(blocks), dim3(256), d, iters);
        t[5] = time_kernel(alu_kernel<5>, dim3(blocks), dim3(256), d, iters);
        t[6] = time_kernel(alu_kernel<6>, dim3(blocks), dim3(256), d, iters);
        t[7] = time_kernel(alu_kernel<7>, dim3(blocks), dim3(256), d, iters);
        for (int m = 0; m < 8; m++) {
            double n_inst = (double)iters * 64;  // wave-instructions per wave
            double waves = (double)blocks * 4;
            // cycles per wave-instruction per SIMD at an assumed 2.4 GHz, all SIMDs busy
            double per_simd = t[m] * 2.4e9 / (n_inst * waves_per_simd);
            printf("waves/SIMD=%d %-16s %.3f ms  %.2f cyc/wave-instr/SIMD  %.2f Tlane-op/s\n", waves_per_simd, names[m], t[m] * 1e3, per_simd,
                   n_inst * waves * 64 / t[m] / 1e12);
        }
    }
    for (int wps = 1; wps <= 8; wps *= 2) {
        int blocks = 256 * wps;
        double tq[3], tr[3];
        tq[0] = time_kernel(fmul_kernel<Fq, 0>, dim3(blocks), dim3(256), (Fq *)d, 200);
        tq[1] = time_kernel(fmul_kernel<Fq, 1>, dim3(blocks), dim3(256), (Fq *)d, 200);
        tq[2] = time_kernel(fmul_kernel<Fq, 2>, dim3(blocks), dim3(256), (Fq *)d, 200);
        tr[0] = time_kernel(fmul_kernel<Fr, 0>, dim3(blocks), dim3(256), (Fr *)d, 200);
        tr[1] = time_kernel(fmul_kernel<Fr, 1>, dim3(blocks), dim3(256), (Fr *)d, 200);
        tr[2] = time_kernel(fmul_kernel<Fr, 2>, dim3(blocks), dim3(256), (Fr *)d, 200);
        const char *vn[3] = {"cios32", "comba1", "comba2"};
        for (int v = 0; v < 3; v++)
            printf("waves/SIMD=%d %s  Fq mul %.2f Gmul/s (%.0f ns per mul per wave)   Fr mul %.2f Gmul/s\n", wps, vn[v],
                   400.0 * blocks * 256 / tq[v] / 1e9, tq[v] / 400 * 1e9, 400.0 * blocks * 256 / tr[v] / 1e9);
    }
    for (int wps = 1; wps <= 8; wps *= 2) {
        int blocks = 256 * wps;
        const double tm = time_kernel(fmul28_kernel<0>, dim3(blocks), dim3(256), (Fq *)d, 200);
        const double ts = time_kernel(fmul28_kernel<1>, dim3(blocks), dim3(256), (Fq *)d, 200);
        printf("waves/SIMD=%d  28-bit limbs: Fq mul %.2f Gmul/s (%.0f ns per mul per wave)   sqr %.2f G/s\n", wps,
               400.0 * blocks * 256 / tm / 1e9, tm / 400 * 1e9, 400.0 * blocks * 256 / ts / 1e9);
    }
    for (int active = 64; active >= 1; active >>= 1) {
        const double tm = time_kernel(fmul28_masked_kernel, dim3(256), dim3(256), (Fq *)d, 200, active);
        printf("1 wave/SIMD, %2d lanes of 64 enabled: %.0f ns per 28-bit product per wave\n", active, tm / 400 * 1e9);
    }
    {
        const unsigned long long masks[] = {0xffffull, 0xfffull, 0x1ffull, 0xffull, 0x00ff00ffull, 0x5555ull, 0x0f0f0f0full, 0x1111111111111111ull,
                                            0x8000000000000001ull, 0x0001000100010001ull, 0x00000000ffff0000ull, 0xff00ull, 0x3ffull, 0x7fffull};
        for (unsigned long long m : masks) {
            const double tm = time_kernel(fmul28_lanemask_kernel, dim3(256), dim3(256), (Fq *)d, 200, m);
            printf("1 wave/SIMD, lane mask %016llx (%2d lanes): %.0f ns per 28-bit product per wave\n", m, __builtin_popcountll(m), tm / 400 * 1e9);
        }
    }
    {
        unsigned long long *dt, ht[512];
        CHECK(hipMalloc(&dt, sizeof ht));
        for (int active = 32; active >= 1; active >>= 1) {
            hipLaunchKernelGGL(fmul28_mixed_kernel, dim3(512), dim3(256), 0, 0, (Fq *)d, 200, active, dt);
            CHECK(hipMemcpy(ht, dt, sizeof ht, hipMemcpyDeviceToHost));
            double full = 0, part = 0;
            for (int b = 0; b < 512; b++) (b & 1 ? part : full) += (double)ht[b];
            printf("mixed launch: blocks with 64 lanes %.0f ns per product, blocks with %2d lanes %.0f ns per product\n", full / 256 * 10 / 400,
                   active, part / 256 * 10 / 400);
        }
        hipFree(dt);
    }
    hipFree(d);
    return 0;
}
