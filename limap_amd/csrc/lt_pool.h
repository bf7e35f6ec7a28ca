// lt_pool.h -- a small persistent team of host threads for the one host pass that matters to the end-to-end time:
// buffering the match rows of a whole scene (lt_triangulate_all_rows).
//
// Why not the OpenMP team: a parallel region whose threads have gone to sleep costs 0.25-0.55 ms to start on the
// 2 x 64-core host of the GPU box (measured: an empty 16-thread region in front of the row pass) -- a quarter of the
// pass -- and OpenMP offers no way to start waking the team before the work is known.  This team can be told in
// advance: wake() returns at once, the workers leave their sleep and spin for a bounded time (kSpinMs); lt_init calls
// it, because in the reference's call sequence the TriangulateImage calls follow Init, so by the time the rows arrive
// the team is running.  A job is one function executed by every worker that is awake (the function shares out the
// work itself through a counter); the calling thread is free between begin() and end() -- lt_triangulate_all_rows uses
// it to enqueue the host -> device copies behind the workers, lt_triangulate_image_rows works along.
#pragma once

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <exception>
#include <cstdio>
#include <mutex>
#include <sched.h>
#include <string>
#include <thread>
#include <type_traits>
#include <unistd.h>
#include <vector>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace lt_host {

// NUMA nodes of the host (round 6).  On the 2 x 64-core host of the GPU box the scheduler spreads a process's threads over
// both sockets; the row pass then runs 1.3-1.9 ms per 100 images, and 0.85-1.0 ms when the process is confined to ONE socket
// (taskset, either socket: the team's shared counters and the caller's arrays stay on one side of the inter-socket link).
// The team therefore FOLLOWS ITS CALLER: a worker that picks up a job binds itself to the CPUs of the node the calling thread
// was running on when it opened the job (one sched_setaffinity when that node changes, never per core -- per-core binding
// was 5 x slower).  The calling thread's own affinity is never touched.  Only CPUs the process is allowed to use are taken;
// hosts with one node, unreadable /sys, or LT_NO_NUMA_FOLLOW=1: no binding.
struct NumaMap {
  std::vector<cpu_set_t> node_set;  // per node: its CPUs that the process may use
  std::vector<int> cpu_node;        // cpu -> node, -1 unknown
  static const NumaMap &get() {
    static const NumaMap m = [] {
      NumaMap r;
      if (getenv("LT_NO_NUMA_FOLLOW")) return r;
      cpu_set_t allowed;
      CPU_ZERO(&allowed);
      if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return r;
      r.cpu_node.assign(CPU_SETSIZE, -1);
      for (int node = 0; node < 64; ++node) {
        const std::string path = "/sys/devices/system/node/node" + std::to_string(node) + "/cpulist";
        FILE *f = std::fopen(path.c_str(), "r");
        if (!f) break;
        char buf[4096];
        const size_t n = std::fread(buf, 1, sizeof(buf) - 1, f);
        std::fclose(f);
        buf[n] = 0;
        cpu_set_t set;
        CPU_ZERO(&set);
        const char *p = buf;
        while (*p) {  // "0-63,128-191"
          char *end = nullptr;
          const long a = std::strtol(p, &end, 10);
          if (end == p) break;
          long b = a;
          p = end;
          if (*p == '-') {
            b = std::strtol(p + 1, &end, 10);
            p = end;
          }
          for (long c = a; c <= b && c < CPU_SETSIZE; ++c)
            if (c >= 0 && CPU_ISSET((int)c, &allowed)) {
              CPU_SET((int)c, &set);
              r.cpu_node[(size_t)c] = node;
            }
          while (*p == ',' || *p == '\n' || *p == ' ') ++p;
        }
        r.node_set.push_back(set);
      }
      int usable = 0;
      for (const cpu_set_t &s : r.node_set) usable += CPU_COUNT(&s) > 0 ? 1 : 0;
      if (usable < 2) r.node_set.clear();  // nothing to choose between
      return r;
    }();
    return m;
  }
  int node_of_caller() const {
    if (node_set.empty()) return -1;
    const int c = sched_getcpu();
    return (c >= 0 && c < (int)cpu_node.size()) ? cpu_node[(size_t)c] : -1;
  }
};

class SpinPool {
 public:
  typedef void (*JobFn)(void *arg, int worker, int n_workers);
  static constexpr double kSpinMs = 2.0;

  // process-wide team, created on first use; a forked child gets its own (threads do not survive a fork)
  static SpinPool &get(int n_workers) {
    static std::mutex mu;
    static SpinPool *pool = nullptr;
    std::lock_guard<std::mutex> lk(mu);
    if (!pool || pool->pid_ != getpid() || pool->n_ != n_workers) pool = new SpinPool(n_workers);  // the old one is left behind
    return *pool;
  }
  int workers() const { return n_; }

  // non-blocking: sleeping workers wake up and spin for kSpinMs waiting for a job
  void wake() {
    want_node_.store(NumaMap::get().node_of_caller(), std::memory_order_relaxed);
    wake_epoch_.fetch_add(1);
    if (sleepers_.load() > 0) notify();
  }
  // A job is open between begin() and end(): workers that are awake (or wake up in time) run fn; end() closes it and
  // waits for the workers inside fn -- it does NOT wait for sleepers, so a caller that arrives while the team sleeps
  // pays one notify and not the team's wake-up time.  The caller therefore needs its own completion condition: it
  // either runs fn itself until the shared work counter is exhausted, or waits on counters fn advances.
  // The team has ONE job slot.  begin() takes it without waiting and returns false when another thread's job holds it
  // (two triangulators on two Python threads: the shim releases the GIL around the blocking calls) or when no worker
  // thread could be created; the caller then runs fn itself and must NOT call end().
  bool begin(JobFn fn, void *arg) {
    if (n_live_ == 0 || !job_mu_.try_lock()) return false;
    want_node_.store(NumaMap::get().node_of_caller(), std::memory_order_relaxed);
    fn_ = fn;
    arg_ = arg;
    const unsigned long long e = job_epoch_.load(std::memory_order_relaxed) + 1;
    open_epoch_.store(e);
    job_epoch_.store(e);  // sequentially consistent with the sleepers count: a worker going to sleep either sees the
    if (sleepers_.load() > 0) notify();  // new epoch in its wait predicate or is counted here and notified
    return true;
  }
  void end() {
    open_epoch_.store(0);  // a worker that has not entered yet stays out (it re-checks after announcing itself)
    while (active_.load() != 0) relax();
    job_mu_.unlock();
  }

 private:
  explicit SpinPool(int n) : n_(n), pid_(getpid()) {
    // a thread that cannot be created (rlimit-constrained container) is simply absent: jobs never wait for a
    // particular worker, and with no worker at all begin() sends the caller to run the job itself
    for (int w = 0; w < n; ++w) {
      try {
        std::thread([this, w] { loop(w); }).detach();
        ++n_live_;
      } catch (...) {
        break;
      }
    }
  }
  static void relax() {
#if defined(__x86_64__)
    _mm_pause();
#endif
  }
  void notify() {
    { std::lock_guard<std::mutex> lk(mu_); }
    cv_.notify_all();
  }
  static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
  }
  void loop(int w) {
    unsigned long long seen_job = 0, seen_wake = 0;
    double deadline = now_ms() + kSpinMs;
    unsigned spins = 0;
    int bound_node = -1;
    auto follow = [&]() {  // onto the caller's NUMA node (see NumaMap)
      const int want = want_node_.load(std::memory_order_relaxed);
      if (want < 0 || want == bound_node) return;
      const NumaMap &nm = NumaMap::get();
      if (want < (int)nm.node_set.size() && CPU_COUNT(&nm.node_set[(size_t)want]) > 0 &&
          sched_setaffinity(0, sizeof(cpu_set_t), &nm.node_set[(size_t)want]) == 0)
        bound_node = want;
    };
    for (;;) {
      const unsigned long long e = job_epoch_.load(std::memory_order_acquire);
      if (e != seen_job) {
        seen_job = e;
        follow();
        active_.fetch_add(1);
        if (open_epoch_.load() == e) {
          try {
            fn_(arg_, w, n_);
          } catch (...) {  // job functions report through their own state (pool_for hands the exception to its caller);
          }                // nothing may unwind through a detached thread
        }
        active_.fetch_sub(1);
        deadline = now_ms() + kSpinMs;
        continue;
      }
      relax();
      if ((++spins & 255u) != 0) continue;
      const unsigned long long wk = wake_epoch_.load(std::memory_order_acquire);
      if (wk != seen_wake) {  // a wake-up call while spinning: spin on from now
        seen_wake = wk;
        follow();
        deadline = now_ms() + kSpinMs;
        continue;
      }
      if (now_ms() < deadline) continue;
      // sleep until a job or a wake-up call arrives
      std::unique_lock<std::mutex> lk(mu_);
      sleepers_.fetch_add(1);
      cv_.wait(lk, [&] { return job_epoch_.load() != seen_job || wake_epoch_.load() != seen_wake; });
      sleepers_.fetch_sub(1);
      seen_wake = wake_epoch_.load();
      lk.unlock();
      follow();
      deadline = now_ms() + kSpinMs;
    }
  }

  const int n_;
  const pid_t pid_;
  int n_live_ = 0;       // worker threads that exist
  std::mutex job_mu_;    // held from begin() to end(): one job at a time
  JobFn fn_ = nullptr;
  void *arg_ = nullptr;
  std::atomic<unsigned long long> job_epoch_{0}, open_epoch_{0}, wake_epoch_{0};
  std::atomic<int> active_{0}, sleepers_{0};
  std::atomic<int> want_node_{-1};  // NUMA node of the thread that last woke the team or opened a job
  std::mutex mu_;
  std::condition_variable cv_;
};

// size of the row-pass team: 15 workers beside the calling thread was the fastest on the 2 x 64-core host (8: +25 %,
// 24-64: slower again -- the pass runs at memory / PCIe speed, more threads only add start-up); LT_ALL_WORKERS overrides
inline int row_workers() {
  static const int n = [] {
    int hw = (int)std::thread::hardware_concurrency();
    int w = hw > 16 ? 15 : (hw > 1 ? hw - 1 : 1);
    if (const char *e = getenv("LT_ALL_WORKERS")) w = std::max(1, std::min(atoi(e), 255));
    return w;
  }();
  return n;
}

// body(begin, end) over [0, n) in pieces of `grain`, shared between the calling thread and the workers that are awake
template <class F>
void pool_for(long long n, long long grain, F &&body) {
  if (n <= 0) return;
  if (n <= grain) {
    body(0ll, n);
    return;
  }
  struct Job {
    std::atomic<long long> next{0};
    long long n, grain;
    typename std::remove_reference<F>::type *f;
    std::mutex err_mu;
    std::exception_ptr err;  // first exception of any piece (e.g. bad_alloc in a worker's scratch): rethrown by the caller
    static void run(void *a, int, int) {
      Job &j = *static_cast<Job *>(a);
      for (;;) {
        const long long b = j.next.fetch_add(j.grain, std::memory_order_relaxed);
        if (b >= j.n) break;
        try {
          (*j.f)(b, std::min(b + j.grain, j.n));
        } catch (...) {
          std::lock_guard<std::mutex> lk(j.err_mu);
          if (!j.err) j.err = std::current_exception();
          j.next.store(j.n, std::memory_order_relaxed);  // the remaining pieces are not started
        }
      }
    }
  } job;
  job.n = n;
  job.grain = grain;
  job.f = &body;
  SpinPool &pool = SpinPool::get(row_workers());
  const bool shared = pool.begin(&Job::run, &job);
  Job::run(&job, 0, 0);
  if (shared) pool.end();
  if (job.err) std::rethrow_exception(job.err);
}

}  // namespace lt_host
