// Host CPU feature probe, compiled WITHOUT the -march=x86-64-v3 -madx flags the rest of the library's host side uses
// (64-bit-limb Montgomery products with mulx / adcx / adox in the MSM folds and G2 line preparation): the library
// refuses to initialise (MI_ERR_NO_DEVICE -> the caller's CPU path) on a host without them instead of dying on SIGILL.
#include <cpuid.h>

extern "C" int mi_host_cpu_ok(void) {
    unsigned a = 0, b = 0, c = 0, d = 0;
    if (!__get_cpuid_count(7, 0, &a, &b, &c, &d)) return 0;
    const int avx2 = (b >> 5) & 1, bmi2 = (b >> 8) & 1, adx = (b >> 19) & 1, bmi1 = (b >> 3) & 1;
    if (!__get_cpuid(1, &a, &b, &c, &d)) return 0;
    const int fma = (c >> 12) & 1, movbe = (c >> 22) & 1, osxsave = (c >> 27) & 1, avx = (c >> 28) & 1;
    return avx2 && bmi1 && bmi2 && adx && fma && movbe && osxsave && avx;
}
