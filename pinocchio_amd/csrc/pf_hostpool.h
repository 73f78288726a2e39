// pf_hostpool.h -- the host threads of the hand-off (pf_get_products, pf_update_products, pf_set_density ...).
//
// Products leave HBM through two pinned staging buffers: while the DMA engine fills one, these threads move the other into the
// caller's (pageable) array -- a plain copy for whole records, a scatter of the named fields for pf_update_products.  Measured on the
// MI355X box of round 6 (profiles/r06_handoff_probe.jsonl, 16 cores of the cgroup): the link gives 55.8 GB/s into pinned memory, one
// hipMemcpy into pageable memory 17 GB/s on first touch and 49 afterwards; sixteen threads copy 134 GB/s and scatter 48 of 104 bytes
// at 57 GB/s of payload -- either keeps up with the link.
#pragma once
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>

#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

// cores this process may use: the affinity mask, cut by the cgroup's CPU quota (cpu.max of cgroup v2, cfs_quota_us of v1) -- on the
// GPU boxes std::thread::hardware_concurrency() says 256 where the cgroup grants 16
static int pf_host_cores() {
  int n = 0;
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
  if (n <= 0) n = (int)std::thread::hardware_concurrency();
  if (n <= 0) n = 1;
  long long quota = -1, period = 0;
  if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char q[64];
    if (fscanf(f, "%63s %lld", q, &period) == 2 && q[0] != 'm') quota = atoll(q);
    fclose(f);
  } else if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
    if (fscanf(g, "%lld", &quota) != 1) quota = -1;
    fclose(g);
    if (FILE *h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(h, "%lld", &period) != 1) period = 0; fclose(h); }
  }
  if (quota > 0 && period > 0) {
    const int c = (int)((quota + period - 1) / period);
    if (c >= 1 && c < n) n = c;
  }
  return n;
}

class PfHostPool {
 public:
  explicit PfHostPool(int nthreads) : n_(nthreads < 1 ? 1 : nthreads) {
    for (int t = 1; t < n_; t++) th_.emplace_back([this, t]() { loop(t); });
  }
  ~PfHostPool() {
    {
      std::lock_guard<std::mutex> l(mu_);
      stop_ = true; gen_++;
    }
    go_.notify_all();
    for (auto &t : th_) t.join();
  }
  int threads() const { return n_; }
  // f(a, b) on n_ disjoint ranges that cover [0, count); returns when all of them are done (the caller works too)
  void run(size_t count, const std::function<void(size_t, size_t)> &f) {
    if (n_ == 1 || count < 4096) { f(0, count); return; }
    {
      std::lock_guard<std::mutex> l(mu_);
      fn_ = &f; count_ = count; pending_ = n_ - 1; gen_++;
    }
    go_.notify_all();
    f(0, count / n_);
    std::unique_lock<std::mutex> l(mu_);
    done_.wait(l, [this]() { return pending_ == 0; });
    fn_ = nullptr;
  }

 private:
  void loop(int t) {
    unsigned long seen = 0;
    for (;;) {
      const std::function<void(size_t, size_t)> *f;
      size_t count;
      {
        std::unique_lock<std::mutex> l(mu_);
        go_.wait(l, [&]() { return gen_ != seen; });
        seen = gen_;
        if (stop_) return;
        f = fn_; count = count_;
      }
      (*f)(count * t / n_, count * (t + 1) / n_);
      {
        std::lock_guard<std::mutex> l(mu_);
        pending_--;
      }
      done_.notify_one();
    }
  }
  int n_;
  std::vector<std::thread> th_;
  std::mutex mu_;
  std::condition_variable go_, done_;
  const std::function<void(size_t, size_t)> *fn_ = nullptr;
  size_t count_ = 0;
  unsigned long gen_ = 0;
  int pending_ = 0;
  bool stop_ = false;
};
