// qdg_pool.hpp -- caching device allocator behind every device buffer of the library.
// Measured on the MI355X pool (round 3): hipMalloc of VRAM that this process has used and freed
// before costs ~34 ms per GiB (the driver scrubs recycled pages) against ~4 ms per GiB for fresh
// memory -- 5 x 13 GiB: 2.2 s against 0.24 s.  A re-mesh (config 5) frees the old mesh and builds a new
// one out of dozens of temporaries, and paid that on every allocation: the device build of an 80.9 M-tet
// mesh took 5.3 s in a process that had run other meshes before, 1.25 s in a fresh one.  Freed blocks
// are therefore kept and handed out again (best fit within 25 % of the request), up to 40 % of the
// device's memory (and at most half of what was free at the process's first allocation); the cache is emptied when an allocation fails (then retried), through
// qdg_device_pool_trim, before the library lets a third party allocate (qdg_comm_create: RCCL) and when the
// process's last context is destroyed (context option "keep_pool" = 1 keeps it: the next context would pay
// the driver again).  Cached bytes are counted per device.  hipFree synchronises the device; a cached block that is handed out again does
// the same (hipDeviceSynchronize) unless it stays on the stream it was freed under (StreamTag below), so a
// buffer freed while kernels may still use it is never reused early.  Allocation never happens inside the time loop.
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstddef>
#include <map>
#include <mutex>
#include <vector>

namespace qdg {

// The stream the calling thread's library call enqueues its work on (set by every entry point that
// allocates or frees device buffers).  A block that was freed under stream S and is handed out again to
// a caller working on S needs no synchronisation: everything the block's previous life enqueued on S runs
// before anything its next life enqueues there.  Any other pairing waits for the device, as hipFree would.
struct StreamTag {
  static hipStream_t& cur() { static thread_local hipStream_t s = nullptr; return s; }
  static bool& known() { static thread_local bool k = false; return k; }
};
struct StreamScope {
  hipStream_t prev; bool prev_known;
  explicit StreamScope(hipStream_t s) : prev(StreamTag::cur()), prev_known(StreamTag::known())
  {
    StreamTag::cur() = s; StreamTag::known() = true;
  }
  ~StreamScope() { StreamTag::cur() = prev; StreamTag::known() = prev_known; }
  StreamScope(const StreamScope&) = delete;
  StreamScope& operator=(const StreamScope&) = delete;
};

class DevicePool {
 public:
  // (never destroyed: buffers may be released during static destruction at process exit)
  static DevicePool& get() { static DevicePool* p = new DevicePool; return *p; }
  hipError_t alloc(void** out, size_t bytes)
  {
    *out = nullptr;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const size_t want = round(bytes);
    bool same_stream = false;
    {
      // a reserved region first
      std::lock_guard<std::mutex> g(mu_);
      for (auto& a : arenas_) {
        if (a.dev != dev) continue;
        auto best = a.free.end();
        for (auto it = a.free.begin(); it != a.free.end(); ++it)
          if (it->second.size >= want && (best == a.free.end() || it->second.size < best->second.size)) best = it;
        if (best == a.free.end()) continue;
        const size_t off = best->first;
        const Range r = best->second;
        a.free.erase(best);
        if (r.size > want) { Range rest = r; rest.size = r.size - want; a.free[off + want] = rest; }
        a.live[a.base + off] = want;
        *out = a.base + off;
        same_stream = r.known && StreamTag::known() && r.stream == StreamTag::cur();
        // (a range nobody has used yet needs no synchronisation either)
        if (r.fresh) same_stream = true;
        break;
      }
    }
    if (*out) return same_stream ? hipSuccess : hipDeviceSynchronize();
    {
      std::lock_guard<std::mutex> g(mu_);
      auto& fl = free_[dev];
      auto it = fl.lower_bound(want);
      if (it != fl.end() && it->first <= want + want / 4) {
        void* p = it->second;
        size_[p] = it->first;
        cached_[dev] -= it->first;
        fl.erase(it);
        auto st = freed_on_.find(p);
        same_stream = st != freed_on_.end() && StreamTag::known() && st->second == StreamTag::cur();
        if (st != freed_on_.end()) freed_on_.erase(st);
        *out = p;
      }
    }
    if (*out) return same_stream ? hipSuccess : hipDeviceSynchronize();   // (what hipFree would have waited for)
    void* p = nullptr;
    e = hipMalloc(&p, want);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      trim();                                       // give the cache back and try once more
      e = hipMalloc(&p, want);
      if (e != hipSuccess) return e;
    }
    std::lock_guard<std::mutex> g(mu_);
    size_[p] = want;
    dev_[p] = dev;
    if (cap_.find(dev) == cap_.end()) {
      size_t fr = 0, tot = 0;
      // 40 % of the device, and no more than half of what was free when this process first asked
      // (several processes may share a GPU in tests)
      cap_[dev] = (hipMemGetInfo(&fr, &tot) == hipSuccess) ? std::min(tot / 5 * 2, fr / 2) : 0;
    }
    *out = p;
    return hipSuccess;
  }
  void free(void* p)
  {
    if (!p) return;
    std::lock_guard<std::mutex> g(mu_);
    for (auto& a : arenas_) {
      auto lv = a.live.find(p);
      if (lv == a.live.end()) continue;
      size_t off = (size_t)(static_cast<char*>(p) - a.base), sz = lv->second;
      a.live.erase(lv);
      Range r{ sz, StreamTag::cur(), StreamTag::known(), false };
      // merge with the free neighbours; a merged range is safe to reuse without a device synchronisation
      // only if all its parts were freed under one stream
      auto nx = a.free.find(off + sz);
      if (nx != a.free.end()) {
        if (!(nx->second.fresh) && !(nx->second.known && r.known && nx->second.stream == r.stream)) r.known = false;
        r.size += nx->second.size;
        a.free.erase(nx);
      }
      auto pv = a.free.lower_bound(off);
      if (pv != a.free.begin()) {
        --pv;
        if (pv->first + pv->second.size == off) {
          if (!(pv->second.fresh) && !(pv->second.known && r.known && pv->second.stream == r.stream)) r.known = false;
          off = pv->first; r.size += pv->second.size;
          a.free.erase(pv);
        }
      }
      a.free[off] = r;
      return;
    }
    auto it = size_.find(p);
    if (it == size_.end()) { (void)hipFree(p); return; }      // not ours
    const int dev = dev_[p];
    const size_t sz = it->second;
    size_.erase(it);
    if (cached_[dev] + sz > cap_[dev]) {                        // this device's cache is full: back to the driver
      dev_.erase(p);
      (void)hipFree(p);
      return;
    }
    free_[dev].emplace(sz, p);
    cached_[dev] += sz;
    if (StreamTag::known()) freed_on_[p] = StreamTag::cur();
  }
  // One region of `bytes` taken from the driver NOW; every later allocation that fits is carved out of it
  // (best fit over its free ranges, neighbours merged on free) and never reaches the driver again -- what
  // a long run needs on this platform, where hipMalloc of memory the process (or an earlier one on the box)
  // has used before costs ~34 ms per GiB: a re-mesh at 80.9 M tets allocates ~100 GB in dozens of pieces.
  hipError_t reserve(size_t bytes)
  {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const size_t want = (bytes + GRAN - 1) / GRAN * GRAN;
    void* p = nullptr;
    e = hipMalloc(&p, want);
    if (e != hipSuccess) { (void)hipGetLastError(); return e; }
    std::lock_guard<std::mutex> g(mu_);
    Arena a;
    a.base = static_cast<char*>(p); a.size = want; a.dev = dev;
    a.free[0] = Range{ want, nullptr, false, true };
    arenas_.push_back(std::move(a));
    return hipSuccess;
  }
  size_t reserved_bytes()
  {
    std::lock_guard<std::mutex> g(mu_);
    size_t n = 0;
    for (auto& a : arenas_) n += a.size;
    return n;
  }
  // hands every cached block back to the driver; returns the bytes released
  size_t trim()
  {
    std::multimap<size_t, void*> all;
    {
      std::lock_guard<std::mutex> g(mu_);
      for (auto& d : free_) { all.insert(d.second.begin(), d.second.end()); d.second.clear(); }
      for (auto& b : all) { dev_.erase(b.second); freed_on_.erase(b.second); }
      cached_.clear();
    }
    size_t n = 0;
    for (auto& b : all) { (void)hipFree(b.second); n += b.first; }
    // reserved regions nothing lives in any more
    std::vector<void*> gone;
    {
      std::lock_guard<std::mutex> g(mu_);
      for (auto it = arenas_.begin(); it != arenas_.end();) {
        if (it->live.empty()) { gone.push_back(it->base); n += it->size; it = arenas_.erase(it); }
        else ++it;
      }
    }
    for (void* p : gone) (void)hipFree(p);
    return n;
  }
  size_t cached_bytes()
  {
    std::lock_guard<std::mutex> g(mu_);
    size_t n = 0;
    for (auto& c : cached_) n += c.second;
    return n;
  }
  // contexts alive in this process: the last one to go returns the cache (qdg_ctx_destroy)
  int ctx_opened() { std::lock_guard<std::mutex> g(mu_); return ++nctx_; }
  int ctx_closed() { std::lock_guard<std::mutex> g(mu_); return nctx_ > 0 ? --nctx_ : 0; }

 private:
  static size_t round(size_t b)
  {
    const size_t q = b >= ((size_t)1 << 21) ? ((size_t)1 << 21) : 4096;   // 2 MiB / 4 KiB granules
    return b == 0 ? q : (b + q - 1) / q * q;
  }
  static constexpr size_t GRAN = (size_t)1 << 21;
  struct Range { size_t size; hipStream_t stream; bool known; bool fresh = true; };
  struct Arena {
    char* base = nullptr; size_t size = 0; int dev = 0;
    std::map<size_t, Range> free;                       // offset -> free range
    std::map<void*, size_t> live;                       // blocks handed out
  };
  std::vector<Arena> arenas_;
  std::mutex mu_;
  std::map<int, std::multimap<size_t, void*>> free_;   // per device: size -> block
  std::map<void*, size_t> size_;                        // live blocks handed out
  std::map<void*, int> dev_;                            // device of every block we own
  std::map<int, size_t> cap_;                           // per device: most bytes kept in the cache
  std::map<int, size_t> cached_;                        // per device: bytes held in free_
  std::map<void*, hipStream_t> freed_on_;               // cached blocks: the stream their last user worked on
  int nctx_ = 0;
};

inline hipError_t dev_alloc(void** p, size_t bytes) { return DevicePool::get().alloc(p, bytes); }
inline void dev_free(void* p) { DevicePool::get().free(p); }

}  // namespace qdg
