// Host-side helper shared by the C ABI (vgt_hip_capi.hip) and the multi-device entry point (vgt_hipx_multi.hip).
#pragma once

#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <thread>
#include <vector>
#ifdef __linux__
#include <sys/mman.h>
#endif

namespace vgt
{
// A large host array that the caller has just allocated (the field a big extraction hands back): its pages do not exist
// yet, and page-locking it faults them in one by one on ONE thread -- 4 GiB: 0.26 - 0.33 s, twice the whole extraction.
// This faults the range in with several threads WITHOUT changing its content, while the call's upload is on its way;
// Wait() before the range is page-locked.  Measured on the MI355X box (tools/microbench/host_pages.cc, 4 GiB): huge pages
// on request + one touch per 2 MiB from 16 threads 19 - 32 ms and 8 ms to page-lock; 4 KiB pages through
// MADV_POPULATE_WRITE 200 ms and 140 ms to page-lock; 4 KiB pages touched one by one from 16 threads 2.3 s (never do that:
// the threads queue on the address space's lock).  So: ask for huge pages, touch the first byte of every 2 MiB, then let
// MADV_POPULATE_WRITE walk the slice -- which finds everything present when the kernel gave huge pages and faults the
// rest in in one call when it did not.
class HostRangePopulator
{
public:
  HostRangePopulator(void* ptr, size_t bytes)
  {
#ifdef __linux__
    if (!ptr || bytes < (size_t{32} << 20)) return;
    constexpr uintptr_t kHuge = uintptr_t{2} << 20;
    const uintptr_t begin = (reinterpret_cast<uintptr_t>(ptr) + 4095) & ~uintptr_t{4095};
    const uintptr_t end = (reinterpret_cast<uintptr_t>(ptr) + bytes) & ~uintptr_t{4095};
    if (end <= begin) return;
    (void)madvise(reinterpret_cast<void*>(begin), end - begin, MADV_HUGEPAGE);
    const unsigned threads = std::max(1u, std::min(16u, std::thread::hardware_concurrency() / 2));
    const uintptr_t slice = (((end - begin) / threads) + kHuge - 1) & ~(kHuge - 1);
    // slices end on absolute 2 MiB boundaries: two threads never fault the same huge page
    for (uintptr_t at = begin; at < end;)
    {
      const uintptr_t stop = std::min<uintptr_t>(end, (at + slice) & ~(kHuge - 1));  // (slice >= 2 MiB: beyond `at`)
      pool_.emplace_back([at, stop]() {
        for (uintptr_t q = at; q < stop; q = (q + kHuge) & ~(kHuge - 1))
        {
          volatile char* c = reinterpret_cast<volatile char*>(q);
          *c = *c;  // (a write fault that keeps the content)
        }
        constexpr int kPopulateWrite = 23;  // MADV_POPULATE_WRITE (Linux 5.14); older kernels: the page-locking faults the rest in
        (void)madvise(reinterpret_cast<void*>(at), stop - at, kPopulateWrite);
      });
      at = stop;
    }
#else
    (void)ptr;
    (void)bytes;
#endif
  }
  void Wait()
  {
    for (auto& th : pool_) th.join();
    pool_.clear();
  }
  ~HostRangePopulator() { Wait(); }
  HostRangePopulator(const HostRangePopulator&) = delete;
  HostRangePopulator& operator=(const HostRangePopulator&) = delete;

private:
  std::vector<std::thread> pool_;
};

}  // namespace vgt
