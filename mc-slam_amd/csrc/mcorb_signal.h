// mcorb_signal.h -- the signal word k_assemble hands the host for every image of a small batch (one rig frame at a time), and the
// checksum it carries.  No HIP dependency: the layout is checked by a plain C++ test (tests/cpp/test_signal.cpp).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define MCORB_SIG_HD __host__ __device__
#else
#define MCORB_SIG_HD
#endif

namespace mcorb {

// The signal word of an image (k_assemble -> host, small batches): bit 0 done, bits 1 - 2 fallback code (1: a level the GPU could not
// select, 2: more than kcap keypoints), bits 3 - 16 keypoint count, bits 17 - 30 monoIndex (both < 16384: kSelSignalMaxCount),
// bits 32 - 63 the XOR of sel_check() over the image's keypoints.
constexpr int kSelSignalMaxCount = 16383;
MCORB_SIG_HD inline unsigned long long sel_signal(int bad, int count, int mono, uint32_t check)
{
    return 1ull | ((unsigned long long)(bad & 3) << 1) | ((unsigned long long)(count & 0x3fff) << 3) | ((unsigned long long)(mono & 0x3fff) << 17) |
           ((unsigned long long)check << 32);
}
MCORB_SIG_HD inline bool sel_signal_done(unsigned long long w) { return (w & 1) != 0; }
MCORB_SIG_HD inline int sel_signal_bad(unsigned long long w) { return (int)((w >> 1) & 3); }
MCORB_SIG_HD inline int sel_signal_count(unsigned long long w) { return (int)((w >> 3) & 0x3fffu); }
MCORB_SIG_HD inline int sel_signal_mono(unsigned long long w) { return (int)((w >> 17) & 0x3fffu); }
MCORB_SIG_HD inline uint32_t sel_signal_check(unsigned long long w) { return (uint32_t)(w >> 32); }
MCORB_SIG_HD inline uint32_t sel_check(uint32_t packed_sel, uint8_t resp, int pos)
{
    return (packed_sel ^ ((uint32_t)resp << 24) ^ ((uint32_t)pos * 0x9E3779B1u)) * 0x85EBCA6Bu;
}

}  // namespace mcorb
