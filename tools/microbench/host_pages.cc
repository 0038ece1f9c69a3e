// What a fresh 4 GiB host array costs before a download can land in it (MI355X box): faulting its pages in (one thread /
// many, with and without huge pages), page-locking it, and the download itself.  hipcc -O2 -o host_pages host_pages.cc
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double Now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void Populate(char* p, size_t bytes, int threads, bool huge, bool populate_call)
{
  if (huge) madvise(p, bytes, MADV_HUGEPAGE);
  std::vector<std::thread> pool;
  const size_t slice = ((bytes / threads) + (2u << 20) - 1) & ~size_t((2u << 20) - 1);
  for (size_t at = 0; at < bytes; at += slice)
  {
    const size_t len = std::min(slice, bytes - at);
    pool.emplace_back([=]() {
      if (populate_call && madvise(p + at, len, 23) == 0) return;
      for (size_t o = 0; o < len; o += 4096) { volatile char* q = p + at + o; *q = *q; }
    });
  }
  for (auto& t : pool) t.join();
}

int main(int argc, char** argv)
{
  const size_t bytes = (argc > 1 ? std::atoll(argv[1]) : 4096ull) << 20;
  void* dev = nullptr;
  hipMalloc(&dev, bytes);
  hipMemset(dev, 1, bytes);
  hipDeviceSynchronize();
  struct Case { const char* name; int threads; bool huge, call, touch; };
  const Case cases[] = {{"register only (fresh pages)", 0, false, false, false},
                        {"1 thread touch", 1, false, false, true},
                        {"16 threads touch", 16, false, false, true},
                        {"16 threads populate", 16, false, true, true},
                        {"16 threads populate, huge", 16, true, true, true},
                        {"16 threads touch, huge", 16, true, false, true},
                        {"32 threads populate", 32, false, true, true},
                        {"64 threads populate", 64, false, true, true}};
  for (const Case& c : cases)
    for (int rep = 0; rep < 2; rep++)
    {
      char* p = static_cast<char*>(mmap(nullptr, bytes + (2u << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0));
      char* a = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(p) + (2u << 20) - 1) & ~uintptr_t((2u << 20) - 1));
      const double t0 = Now();
      if (c.touch) Populate(a, bytes, c.threads, c.huge, c.call);
      const double t1 = Now();
      const hipError_t e = hipHostRegister(a, bytes, hipHostRegisterDefault);
      const double t2 = Now();
      hipMemcpy(a, dev, bytes, hipMemcpyDeviceToHost);
      const double t3 = Now();
      if (e == hipSuccess) hipHostUnregister(a);
      const double t4 = Now();
      munmap(p, bytes + (2u << 20));
      const double t5 = Now();
      std::printf("%-32s populate %7.1f  register %7.1f (%d)  download %7.1f  unregister %6.1f  munmap %6.1f ms\n", c.name,
                  (t1 - t0) * 1e3, (t2 - t1) * 1e3, int(e), (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3);
    }
  // present pages (reused array): register + download
  char* a = static_cast<char*>(aligned_alloc(2u << 20, bytes));
  std::memset(a, 0, bytes);
  for (int rep = 0; rep < 2; rep++)
  {
    const double t1 = Now();
    hipHostRegister(a, bytes, hipHostRegisterDefault);
    const double t2 = Now();
    hipMemcpy(a, dev, bytes, hipMemcpyDeviceToHost);
    const double t3 = Now();
    hipHostUnregister(a);
    const double t4 = Now();
    std::printf("%-32s register %7.1f  download %7.1f  unregister %6.1f ms\n", "reused array", (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3);
  }
  std::printf("hardware threads %u\n", std::thread::hardware_concurrency());
  FILE* f = std::fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r");
  if (f) { char line[128] = {0}; if (std::fgets(line, 127, f)) std::printf("thp enabled: %s", line); std::fclose(f); }
  f = std::fopen("/sys/kernel/mm/transparent_hugepage/defrag", "r");
  if (f) { char line[128] = {0}; if (std::fgets(line, 127, f)) std::printf("thp defrag: %s", line); std::fclose(f); }
  return 0;
}
