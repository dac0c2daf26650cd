// Launch tape: a recorded multi-stream replay of one train step.
//
// The step is ~800 kernel launches sequenced by Python over four HIP streams (main, language, caption, weight-gradient).
// Issuing them from Python costs ~13 us each (10.5 ms per step, more than the GPU needs).  hipGraph capture removes the
// host cost but, on this ROCm, replays the forked branches serially.  The tape keeps both: while `recording`, every
// L2S_LAUNCH site (common.h) also appends a closure {logical stream, kernel, launch geometry, argument copies} and every
// l2s_stream_fork appends an event record/wait pair; l2s_tape_run replays the closures from one tight C++ loop onto the
// caller's streams, so the branches still overlap on the GPU.  All arguments are persistent device pointers / sizes, so a
// tape stays valid for as long as the shapes (and scalar hyper-parameters) it was recorded with.
#include "common.h"
#include "../../include/lang2seg_hip.h"
#include <vector>
#include <mutex>

namespace l2s {

struct Op { int kind; int sid; int ev; std::function<void(hipStream_t)> fn; };   // kind 0 launch, 1 record ev, 2 wait ev, 3 segment mark, 4 record timing ev, 5 record slot, 6 wait slot
struct Tape { std::vector<Op> ops; int n_events = 0; std::vector<hipEvent_t> events; std::vector<size_t> marks; };

static Tape* g_rec = nullptr;
bool g_paused = false;   // l2s_tape_pause: launches issued by the host between two segments (the gradient reducer's casts) are not recorded
static hipStream_t g_streams[8];
static int g_nstreams = 0;
static std::vector<hipEvent_t> g_pool;     // events for eager forks
static size_t g_pool_next = 0;
static std::vector<hipEvent_t> g_timing;  // timing events of l2s_tape_time_event (measurement only; live until the process ends)
// Event slots: process-wide named points of a stream's order.  An ordinary fork is "record now, wait now"; a slot separates the two, so that
// a point recorded at the END of one step (by one tape) can be waited for in the MIDDLE of the next (by another tape): the first part of
// the optimiser update is awaited before layer2 while the deferred weight gradients keep running on the same stream behind it.
static const int kSlots = 16;
static hipEvent_t g_slot[kSlots];
static bool g_slot_made[kSlots];
static hipEvent_t slot_event(int slot) {
  if (slot < 0 || slot >= kSlots) return nullptr;
  if (!g_slot_made[slot]) {
    if (hipEventCreateWithFlags(&g_slot[slot], hipEventDisableTiming) != hipSuccess) return nullptr;
    g_slot_made[slot] = true;
  }
  return g_slot[slot];
}

static int stream_id(hipStream_t s) {
  for (int i = 0; i < g_nstreams; ++i) if (g_streams[i] == s) return i;
  return -1;
}
bool recording() { return g_rec != nullptr && !g_paused; }
void record(hipStream_t s, std::function<void(hipStream_t)> fn) {
  if (!g_rec || g_paused) return;
  g_rec->ops.push_back(Op{0, stream_id(s), -1, std::move(fn)});
}

}  // namespace l2s

using namespace l2s;

extern "C" void* l2s_tape_begin(const hipStream_t* streams, int n) {
  if (g_rec || n < 1 || n > 8) return nullptr;
  for (int i = 0; i < n; ++i) g_streams[i] = streams[i];
  g_nstreams = n;
  g_rec = new Tape();
  return g_rec;
}
extern "C" int l2s_tape_end(void* tape) {
  if (g_rec != (Tape*)tape || !tape) return L2S_EINVAL;
  Tape* t = g_rec;
  g_rec = nullptr;
  g_paused = false;
  for (size_t i = 0; i < t->ops.size(); ++i) {
    if (t->ops[i].kind == 3) t->marks.push_back(i);
    else if (t->ops[i].sid < 0) return L2S_EINVAL;                  // a launch went to an unregistered stream
  }
  t->events.resize(t->n_events);
  for (int i = 0; i < t->n_events; ++i)
    if (hipEventCreateWithFlags(&t->events[i], hipEventDisableTiming) != hipSuccess) return L2S_ELAUNCH;
  return L2S_OK;
}
extern "C" long l2s_tape_size(void* tape) { return tape ? (long)((Tape*)tape)->ops.size() : -1; }
// Segments: l2s_tape_mark() splits a tape at the points where the host has to act between launches (the data-parallel
// gradient all-reduce of a finished bucket goes through torch.distributed / RCCL, which cannot be recorded): segment k is
// everything between mark k-1 and mark k.
// While paused, launches execute but are not recorded: the data-parallel reducer runs between two segments on its own stream (its
// collectives cannot be recorded, and its pack / unpack launches must not be replayed by the tape AND issued again by the reducer).
extern "C" int l2s_tape_pause(int on) { g_paused = on != 0; return L2S_OK; }
extern "C" int l2s_tape_mark(void) {
  if (g_rec) g_rec->ops.push_back(Op{3, 0, -1, nullptr});
  return L2S_OK;
}
// Timing events on the tape (measurement): while recording, l2s_tape_time_event(s) appends "record timing event #id on s" at this point of
// the stream's order and returns id (>= 0); every replay records the event again, so after a synchronise l2s_time_event_elapsed(a, b)
// is the HIP-event time between two points of the LAST replayed step — launches inside the pipelined, host-unbound replay can be
// bracketed this way (events around eager launches also measure the host's issue gaps).  Outside a recording it returns -1.
extern "C" int l2s_tape_time_event(hipStream_t s) {
  if (!g_rec) return -1;
  const int sid = stream_id(s);
  if (sid < 0) return -1;
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return -1;
  g_timing.push_back(e);
  const int id = (int)g_timing.size() - 1;
  g_rec->ops.push_back(Op{4, sid, id, nullptr});
  (void)hipEventRecord(e, s);                                    // the recording step executes as well
  return id;
}
extern "C" int l2s_time_event_elapsed(int a, int b, float* ms) {
  if (!ms || a < 0 || b < 0 || a >= (int)g_timing.size() || b >= (int)g_timing.size()) return L2S_EINVAL;
  return hipEventElapsedTime(ms, g_timing[a], g_timing[b]) == hipSuccess ? L2S_OK : L2S_ELAUNCH;
}
extern "C" int l2s_tape_segments(void* tape) { return tape ? (int)((Tape*)tape)->marks.size() + 1 : -1; }
extern "C" int l2s_tape_run_segment(void* tape, const hipStream_t* streams, int n, int seg) {
  Tape* t = (Tape*)tape;
  if (!t || g_rec || seg < 0 || seg > (int)t->marks.size()) return L2S_EINVAL;
  const size_t lo = seg == 0 ? 0 : t->marks[seg - 1] + 1, hi = seg == (int)t->marks.size() ? t->ops.size() : t->marks[seg];
  for (size_t i = lo; i < hi; ++i) {
    const Op& o = t->ops[i];
    if (o.sid >= n) return L2S_EINVAL;
    hipStream_t s = streams[o.sid];
    if (o.kind == 0) o.fn(s);
    else if (o.kind == 1) { if (hipEventRecord(t->events[o.ev], s) != hipSuccess) return L2S_ELAUNCH; }
    else if (o.kind == 2) { if (hipStreamWaitEvent(s, t->events[o.ev], 0) != hipSuccess) return L2S_ELAUNCH; }
    else if (o.kind == 4) { if (hipEventRecord(g_timing[o.ev], s) != hipSuccess) return L2S_ELAUNCH; }
    else if (o.kind == 5) { if (hipEventRecord(g_slot[o.ev], s) != hipSuccess) return L2S_ELAUNCH; }
    else if (o.kind == 6) { if (hipStreamWaitEvent(s, g_slot[o.ev], 0) != hipSuccess) return L2S_ELAUNCH; }
  }
  return l2s_check_launch();
}
extern "C" int l2s_tape_run(void* tape, const hipStream_t* streams, int n) {
  Tape* t = (Tape*)tape;
  if (!t || g_rec) return L2S_EINVAL;
  for (const Op& o : t->ops) {
    if (o.kind == 3) continue;
    if (o.sid >= n) return L2S_EINVAL;
    hipStream_t s = streams[o.sid];
    if (o.kind == 0) o.fn(s);
    else if (o.kind == 1) { if (hipEventRecord(t->events[o.ev], s) != hipSuccess) return L2S_ELAUNCH; }
    else if (o.kind == 2) { if (hipStreamWaitEvent(s, t->events[o.ev], 0) != hipSuccess) return L2S_ELAUNCH; }
    else if (o.kind == 4) { if (hipEventRecord(g_timing[o.ev], s) != hipSuccess) return L2S_ELAUNCH; }
    else if (o.kind == 5) { if (hipEventRecord(g_slot[o.ev], s) != hipSuccess) return L2S_ELAUNCH; }
    else if (o.kind == 6) { if (hipStreamWaitEvent(s, g_slot[o.ev], 0) != hipSuccess) return L2S_ELAUNCH; }
  }
  return l2s_check_launch();
}
extern "C" int l2s_tape_destroy(void* tape) {
  Tape* t = (Tape*)tape;
  if (!t) return L2S_OK;
  if (g_rec == t) g_rec = nullptr;
  for (hipEvent_t e : t->events) hipEventDestroy(e);
  delete t;
  return L2S_OK;
}
// `to` waits for everything enqueued so far on `from` (fork or join, depending on which side continues)
extern "C" int l2s_stream_fork(hipStream_t from, hipStream_t to) {
  if (from == to) return L2S_OK;
  if (g_pool.empty()) {
    g_pool.resize(256);
    for (auto& e : g_pool) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return L2S_ELAUNCH;
  }
  hipEvent_t e = g_pool[g_pool_next++ % g_pool.size()];
  if (hipEventRecord(e, from) != hipSuccess || hipStreamWaitEvent(to, e, 0) != hipSuccess) return L2S_ELAUNCH;
  if (g_rec) {
    const int a = stream_id(from), b = stream_id(to);
    const int ev = g_rec->n_events++;
    g_rec->ops.push_back(Op{1, a, ev, nullptr});
    g_rec->ops.push_back(Op{2, b, ev, nullptr});
  }
  return L2S_OK;
}
// slot: record "everything enqueued so far on s" under a process-wide name / make s wait for the most recent record of that name
// (a slot that was never recorded is complete: the wait is a no-op, as for any HIP event)
extern "C" int l2s_event_record(int slot, hipStream_t s) {
  hipEvent_t e = slot_event(slot);
  if (!e) return L2S_EINVAL;
  if (hipEventRecord(e, s) != hipSuccess) return L2S_ELAUNCH;
  if (g_rec && !g_paused) g_rec->ops.push_back(Op{5, stream_id(s), slot, nullptr});
  return L2S_OK;
}
extern "C" int l2s_event_wait(int slot, hipStream_t s) {
  hipEvent_t e = slot_event(slot);
  if (!e) return L2S_EINVAL;
  if (hipStreamWaitEvent(s, e, 0) != hipSuccess) return L2S_ELAUNCH;
  if (g_rec && !g_paused) g_rec->ops.push_back(Op{6, stream_id(s), slot, nullptr});
  return L2S_OK;
}
// Clears are ordinary kernel launches (recorded on the tape like every other launch; word-aligned ranges only).
__global__ __launch_bounds__(256) void fill_words_kernel(uint32_t* p, size_t nwords, uint32_t v) {
  size_t head = ((16 - ((uintptr_t)p & 15)) & 15) >> 2;
  if (head > nwords) head = nwords;
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, nthreads = gridDim.x * (size_t)blockDim.x;
  if (i < head) p[i] = v;
  uint4* q = (uint4*)(p + head);
  const size_t nq = (nwords - head) >> 2;
  for (size_t j = i; j < nq; j += nthreads) q[j] = make_uint4(v, v, v, v);
  const size_t tail0 = head + (nq << 2);
  if (i < nwords - tail0) p[tail0 + i] = v;
}
extern "C" int l2s_memset_async(void* p, int value, size_t bytes, hipStream_t s) {
  if (bytes == 0) return L2S_OK;
  if (((uintptr_t)p & 3) || (bytes & 3)) {
    // byte-granular clears (none on the train-step path) stay with the runtime
    if (recording()) record(s, [=](hipStream_t s_) { (void)hipMemsetAsync(p, value, bytes, s_); });
    return hipMemsetAsync(p, value, bytes, s) == hipSuccess ? L2S_OK : L2S_ELAUNCH;
  }
  const uint32_t b = (uint32_t)value & 0xffu, v = b | (b << 8) | (b << 16) | (b << 24);
  const size_t nwords = bytes >> 2;
  size_t blocks = (nwords / 4 + 255) / 256;                 // one 16-byte store per thread up to 8192 workgroups, then a grid-stride loop
  if (blocks < 1) blocks = 1;
  if (blocks > 8192) blocks = 8192;
  L2S_LAUNCH(fill_words_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (uint32_t*)p, nwords, v);
  return l2s_check_launch();
}
extern "C" int l2s_memcpy_d2d_async(void* dst, const void* src, size_t bytes, hipStream_t s) {
  if (bytes == 0) return L2S_OK;
  if (recording()) record(s, [=](hipStream_t s_) { (void)hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s_); });
  return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s) == hipSuccess ? L2S_OK : L2S_ELAUNCH;
}
