// Deferred finishing of parameter-gradient sums (ptpp_common.h: red_take / red_finish; include/ptpp.h "Deferred reduction").
//
// The cross-block column sums of this library go through 32 replicated scratch rows and a one-wave-per-64-columns finishing
// launch (ptpp_common.h).  For a PARAMETER gradient nobody reads the total before the optimiser, so the finishing launch can wait:
// every stream that produces such sums owns a sub-arena; a producer takes a zeroed slice, its (slice, destination) pair is queued,
// and ptpp_red_flush() finishes all queued sums of a stream with ONE launch ON THAT STREAM (the kernel zeroes the slices again) and
// makes the caller's stream wait for it.  Producer, finisher and the next producer that reuses a slice are all on the same stream,
// so no cross-stream ordering is assumed.  Host state is guarded by a mutex (autograd may call from a worker thread).
#include <mutex>
#include <vector>

#include "ptpp_common.h"

namespace {

struct RedEntry {
  float* scratch;
  float* dst0;
  float* dst1;
  int KC, n0, accumulate, blk0;
};
constexpr int RED_MAX_BATCH = 48;
struct RedBatch {
  RedEntry e[RED_MAX_BATCH];
  int n;
};

__global__ __launch_bounds__(64) void red_sum_batched_kernel(const RedBatch tab) {
  int i = 0;
  while (i + 1 < tab.n && (int)blockIdx.x >= tab.e[i + 1].blk0) ++i;
  const RedEntry en = tab.e[i];
  const int c = ((int)blockIdx.x - en.blk0) * 64 + threadIdx.x;
  if (c >= en.KC) return;
  float v[PTPP_RED_NREP];
#pragma unroll
  for (int r = 0; r < PTPP_RED_NREP; ++r) v[r] = en.scratch[(size_t)r * en.KC + c];
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < PTPP_RED_NREP; ++r) {
    s += v[r];
    en.scratch[(size_t)r * en.KC + c] = 0.f;
  }
  float* d = c < en.n0 ? (en.dst0 ? en.dst0 + c : nullptr) : (en.dst1 ? en.dst1 + (c - en.n0) : nullptr);
  // (atomic: several queued sums may share a destination -- a parameter used twice in the forward -- and run in one launch)
  if (d) {
    if (en.accumulate) unsafeAtomicAdd(d, s);
    else *d = s;
  }
}

struct StreamArena {
  hipStream_t st;
  char* base;
  size_t used;
  hipEvent_t ev;
  std::vector<RedEntry> pending;
};
struct RedState {
  std::mutex mu;
  char* arena = nullptr;
  size_t bytes = 0, sub = 0;
  int nsub = 0, suspended = 0;
  std::vector<StreamArena> streams;
};
RedState& state() {
  static RedState s;
  return s;
}
constexpr int RED_MAX_STREAMS = 4;

}  // namespace

void* ptpp_red_arena_take(size_t bytes, hipStream_t st) {
  RedState& S = state();
  std::lock_guard<std::mutex> g(S.mu);
  if (!S.arena || S.suspended > 0) return nullptr;
  StreamArena* sa = nullptr;
  for (auto& a : S.streams)
    if (a.st == st) sa = &a;
  if (!sa) {
    if ((int)S.streams.size() >= S.nsub) return nullptr;
    StreamArena a;
    a.st = st;
    a.base = S.arena + S.streams.size() * S.sub;
    a.used = 0;
    if (hipEventCreateWithFlags(&a.ev, hipEventDisableTiming) != hipSuccess) return nullptr;
    S.streams.push_back(a);
    sa = &S.streams.back();
  }
  bytes = (bytes + 255) & ~(size_t)255;
  if (sa->used + bytes > S.sub) return nullptr;
  void* p = sa->base + sa->used;
  sa->used += bytes;
  return p;
}

void ptpp_red_arena_push(void* slice, int KC, float* dst0, int n0, float* dst1, int accumulate, hipStream_t st) {
  RedState& S = state();
  std::lock_guard<std::mutex> g(S.mu);
  for (auto& a : S.streams)
    if (a.st == st) {
      a.pending.push_back(RedEntry{reinterpret_cast<float*>(slice), dst0, dst1, KC, n0, accumulate, 0});
      return;
    }
}

extern "C" int ptpp_red_defer(void* arena, size_t bytes) {
  RedState& S = state();
  std::lock_guard<std::mutex> g(S.mu);
  for (auto& a : S.streams)
    PTPP_CHECK_ARG(a.pending.empty(), "red_defer: %zu sums are still queued (ptpp_red_flush first)", a.pending.size());
  for (auto& a : S.streams) (void)hipEventDestroy(a.ev);
  S.streams.clear();
  S.arena = reinterpret_cast<char*>(arena);
  S.bytes = arena ? bytes : 0;
  S.sub = (S.bytes / RED_MAX_STREAMS) & ~(size_t)255;
  S.nsub = S.sub >= (size_t)(1 << 20) ? RED_MAX_STREAMS : 0;
  PTPP_CHECK_ARG(!arena || S.nsub > 0, "red_defer: the arena must hold at least %d MiB", RED_MAX_STREAMS);
  return PTPP_OK;
}

extern "C" int ptpp_red_defer_suspend(int delta) {
  RedState& S = state();
  std::lock_guard<std::mutex> g(S.mu);
  S.suspended += delta;
  if (S.suspended < 0) S.suspended = 0;
  return PTPP_OK;
}

extern "C" int ptpp_red_pending(void) {
  RedState& S = state();
  std::lock_guard<std::mutex> g(S.mu);
  size_t n = 0;
  for (auto& a : S.streams) n += a.pending.size();
  return (int)n;
}

extern "C" int ptpp_red_flush(void* main_stream) {
  RedState& S = state();
  std::lock_guard<std::mutex> g(S.mu);
  hipStream_t main = reinterpret_cast<hipStream_t>(main_stream);
  for (auto& a : S.streams) {
    if (a.pending.empty()) continue;
    size_t i = 0;
    while (i < a.pending.size()) {
      RedBatch tab;
      tab.n = 0;
      int blk = 0;
      for (; i < a.pending.size() && tab.n < RED_MAX_BATCH; ++i) {
        RedEntry e = a.pending[i];
        e.blk0 = blk;
        blk += (e.KC + 63) / 64;
        tab.e[tab.n++] = e;
      }
      hipLaunchKernelGGL(red_sum_batched_kernel, dim3((unsigned)blk), dim3(64), 0, a.st, tab);
    }
    a.pending.clear();
    a.used = 0;  // the launch above re-zeroed every slice; the next producer on this stream is ordered behind it
    if (a.st != main) {
      if (hipEventRecord(a.ev, a.st) != hipSuccess || hipStreamWaitEvent(main, a.ev, 0) != hipSuccess) {
        ptpp_set_error("red_flush: joining a producer stream failed");
        return PTPP_ELAUNCH;
      }
    }
  }
  PTPP_CHECK_LAUNCH("red_flush");
  return PTPP_OK;
}
