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
#include <mutex>
#include <thread>
#include <type_traits>
#include <unistd.h>
#include <vector>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace lt_host {

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
    for (;;) {
      const unsigned long long e = job_epoch_.load(std::memory_order_acquire);
      if (e != seen_job) {
        seen_job = e;
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
