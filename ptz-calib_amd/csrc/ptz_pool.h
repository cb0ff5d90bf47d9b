// ptz_pool.h -- process-wide cache of device allocations, pinned host blocks, streams and events.
// The reference constructs a fresh optimizer object per solve (one ceres::Problem per PTZRayOptimizer / KRTOptimizer,
// ptzray_optimizer.h:145-149), and PTZ-IBA does that ~2 N times for an N-view rig.  On the device a fresh set of
// hipMalloc / hipStreamCreate / hipEventCreate calls per solve costs milliseconds -- more than the solve itself for small
// problems -- so released resources are parked here and handed to the next solve instead of going back to the driver.
//  * device blocks: size classes {1, 1.25, 1.5, 1.75} x 2^k (<= 25 % slack), parked up to a byte budget
//    (PTZ_CACHE_MAX_MB, default 32768); anything beyond the budget is freed immediately, so large batches do not pin HBM;
//  * everything is released by ptz_trim_cache().
#pragma once

#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

namespace ptzpool {

struct State {
  std::mutex mu;
  struct PerDev {
    std::map<size_t, std::vector<void*>> free_blocks;   // size class -> parked blocks
    std::unordered_map<void*, size_t> live;             // block -> size class
    size_t parked_bytes = 0;
    std::vector<hipStream_t> streams;
    std::vector<hipEvent_t> timing_events, plain_events;
  };
  std::map<int, PerDev> dev;
  std::map<size_t, std::vector<void*>> pinned_free;
  std::unordered_map<void*, size_t> pinned_live;
  size_t budget = 0;
  State()  // (in the constructor: the function-local static below is initialised exactly once, also under concurrent first use)
  {
    size_t mb = 32768;  // of 288 GB: successive batches of different size (the lock-step PTZ-IBA) otherwise thrash hipMalloc / hipFree
    if (const char* e = getenv("PTZ_CACHE_MAX_MB")) mb = (size_t)atoll(e);
    budget = mb << 20;
  }
};

inline State& state()
{
  static State s;
  return s;
}

// Debug aid: PTZ_POOL_TRACE=1 reports every request the cache could not serve (the driver call it took, and how long) on stderr.
inline bool trace_on()
{
  static const bool on = getenv("PTZ_POOL_TRACE") != nullptr;
  return on;
}
struct MissTimer {
  const char* what; size_t bytes; std::chrono::steady_clock::time_point t0;
  MissTimer(const char* w, size_t b) : what(w), bytes(b) { if (trace_on()) t0 = std::chrono::steady_clock::now(); }
  ~MissTimer()
  {
    if (trace_on())
      fprintf(stderr, "ptzpool miss %s %zu bytes %.3f ms\n", what, bytes,
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  }
};

inline size_t size_class(size_t bytes)
{
  if (bytes < 256) return 256;
  size_t p = 256;
  while (p * 2 <= bytes) p *= 2;  // p <= bytes < 2p
  if (bytes == p) return p;
  const size_t q = p / 4;
  return p + ((bytes - p + q - 1) / q) * q;
}

// Debug aid: PTZ_POOL_FILL=<0..255> fills every block handed out with that byte (0xFF = NaN doubles), so that a kernel
// reading memory it was not given initialised shows up as NaNs instead of depending on what the last solve left behind.
inline int fill_byte()
{
  const char* e = getenv("PTZ_POOL_FILL");  // read per call so that tests can switch it inside one process
  return e ? atoi(e) & 255 : -1;
}

inline hipError_t dev_acquire_raw(int device, size_t bytes, void** out);
inline hipError_t dev_acquire(int device, size_t bytes, void** out)
{
  const hipError_t e = dev_acquire_raw(device, bytes, out);
  if (e == hipSuccess && fill_byte() >= 0) (void)hipMemset(*out, fill_byte(), size_class(bytes));
  return e;
}

inline hipError_t dev_acquire_raw(int device, size_t bytes, void** out)
{
  State& s = state();
  const size_t cls = size_class(bytes);
  {
    std::lock_guard<std::mutex> lk(s.mu);
    auto& d = s.dev[device];
    auto it = d.free_blocks.find(cls);
    if (it != d.free_blocks.end() && !it->second.empty()) {
      *out = it->second.back();
      it->second.pop_back();
      d.parked_bytes -= cls;
      d.live[*out] = cls;
      return hipSuccess;
    }
  }
  MissTimer mt("hipMalloc", cls);
  hipError_t e = hipMalloc(out, cls);
  if (e != hipSuccess) {
    // the cache may be what is in the way: give everything parked back and retry once
    {
      std::lock_guard<std::mutex> lk(s.mu);
      auto& d = s.dev[device];
      for (auto& kv : d.free_blocks)
        for (void* p : kv.second) (void)hipFree(p);
      d.free_blocks.clear();
      d.parked_bytes = 0;
    }
    (void)hipGetLastError();
    e = hipMalloc(out, cls);
    if (e != hipSuccess) return e;
  }
  std::lock_guard<std::mutex> lk(s.mu);
  s.dev[device].live[*out] = cls;
  return hipSuccess;
}

inline void dev_release(int device, void* p)
{
  if (!p) return;
  State& s = state();
  size_t cls = 0;
  bool park = false;
  {
    std::lock_guard<std::mutex> lk(s.mu);
    auto& d = s.dev[device];
    auto it = d.live.find(p);
    if (it == d.live.end()) { cls = 0; }
    else {
      cls = it->second;
      d.live.erase(it);
      if (d.parked_bytes + cls <= s.budget) {
        d.free_blocks[cls].push_back(p);
        d.parked_bytes += cls;
        park = true;
      }
    }
  }
  if (!park) {
    MissTimer mt("hipFree", cls);
    (void)hipFree(p);
  }
}

inline hipError_t pinned_acquire(size_t bytes, void** out)
{
  State& s = state();
  const size_t cls = size_class(bytes);
  {
    std::lock_guard<std::mutex> lk(s.mu);
    auto it = s.pinned_free.find(cls);
    if (it != s.pinned_free.end() && !it->second.empty()) {
      *out = it->second.back();
      it->second.pop_back();
      s.pinned_live[*out] = cls;
      return hipSuccess;
    }
  }
  MissTimer mt("hipHostMalloc", cls);
  hipError_t e = hipHostMalloc(out, cls);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lk(s.mu);
  s.pinned_live[*out] = cls;
  return hipSuccess;
}

inline void pinned_release(void* p)
{
  if (!p) return;
  State& s = state();
  std::lock_guard<std::mutex> lk(s.mu);
  auto it = s.pinned_live.find(p);
  if (it == s.pinned_live.end()) return;
  s.pinned_free[it->second].push_back(p);
  s.pinned_live.erase(it);
}

inline hipError_t stream_acquire(int device, hipStream_t* out)
{
  State& s = state();
  {
    std::lock_guard<std::mutex> lk(s.mu);
    auto& v = s.dev[device].streams;
    if (!v.empty()) { *out = v.back(); v.pop_back(); return hipSuccess; }
  }
  MissTimer mt("hipStreamCreate", 0);
  return hipStreamCreate(out);
}

inline void stream_release(int device, hipStream_t st)
{
  if (!st) return;
  State& s = state();
  std::lock_guard<std::mutex> lk(s.mu);
  s.dev[device].streams.push_back(st);
}

inline hipError_t event_acquire(int device, bool timing, hipEvent_t* out)
{
  State& s = state();
  {
    std::lock_guard<std::mutex> lk(s.mu);
    auto& v = timing ? s.dev[device].timing_events : s.dev[device].plain_events;
    if (!v.empty()) { *out = v.back(); v.pop_back(); return hipSuccess; }
  }
  MissTimer mt("hipEventCreate", 0);
  return timing ? hipEventCreate(out) : hipEventCreateWithFlags(out, hipEventDisableTiming);
}

inline void event_release(int device, bool timing, hipEvent_t e)
{
  if (!e) return;
  State& s = state();
  std::lock_guard<std::mutex> lk(s.mu);
  (timing ? s.dev[device].timing_events : s.dev[device].plain_events).push_back(e);
}

// Give every parked resource back to the driver (resources in use by live batches are not touched).
inline void trim()
{
  // handles are collected under the lock and given back outside it: hipFree / hipStreamDestroy can synchronise the device,
  // which must not happen while another thread's create waits for the pool
  struct Freed { int dev; std::vector<void*> blocks; std::vector<hipStream_t> streams; std::vector<hipEvent_t> events; };
  std::vector<Freed> freed;
  std::vector<void*> pinned;
  {
    State& s = state();
    std::lock_guard<std::mutex> lk(s.mu);
    for (auto& kv : s.dev) {
      Freed f;
      f.dev = kv.first;
      for (auto& fb : kv.second.free_blocks) f.blocks.insert(f.blocks.end(), fb.second.begin(), fb.second.end());
      kv.second.free_blocks.clear();
      kv.second.parked_bytes = 0;
      f.streams.swap(kv.second.streams);
      f.events.swap(kv.second.timing_events);
      f.events.insert(f.events.end(), kv.second.plain_events.begin(), kv.second.plain_events.end());
      kv.second.plain_events.clear();
      freed.push_back(std::move(f));
    }
    for (auto& kv : s.pinned_free) pinned.insert(pinned.end(), kv.second.begin(), kv.second.end());
    s.pinned_free.clear();
  }
  int cur = 0;
  (void)hipGetDevice(&cur);
  for (auto& f : freed) {
    (void)hipSetDevice(f.dev);
    for (void* p : f.blocks) (void)hipFree(p);
    for (auto st : f.streams) (void)hipStreamDestroy(st);
    for (auto e : f.events) (void)hipEventDestroy(e);
  }
  for (void* p : pinned) (void)hipHostFree(p);
  (void)hipSetDevice(cur);
}

}  // namespace ptzpool
