// Native launch tape: the step's static launch sequence (C-ABI calls with their arguments, event records, stream waits)
// recorded once by the host executor and re-issued from C without the interpreter (the executor seam; the
// reference issues the same sequence from Python / autograd every step, train.py:2743-2767).
//
// A tape is a flat list of operations; az_tape_play issues them in order until it meets a BREAK (host logic that must run in
// the caller: data-parallel region waits, scheduling hints) or the end, and returns the index to resume from.  Calls are
// dispatched through a switch generated from include/aozora_hip.h (az_tape_dispatch.inc): arguments travel as 64-bit words
// (int / long by value, float by bit pattern, pointers as integers).
#include "az_common.h"
#include "aozora_hip.h"
#include <cstring>
#include <vector>

namespace {
inline float word_float(long w) { float f; unsigned u = (unsigned)(unsigned long)w; memcpy(&f, &u, 4); return f; }
#include "az_tape_dispatch.inc"
constexpr int AZ_TAPE_NFN = (int)(sizeof(AZ_TAPE_FN_NAMES) / sizeof(AZ_TAPE_FN_NAMES[0]));
constexpr int MAXW = 40;
enum { OP_CALL = 0, OP_EVENT_RECORD = 1, OP_STREAM_WAIT = 2, OP_BREAK = 3 };
struct Op { int kind, fn, nw; long w[MAXW]; };
struct Tape { std::vector<Op> ops; long err_index = -1; int err_rc = 0; };
}  // namespace

extern "C" {

int az_tape_fn_id(const char* name) {
  if (!name) return -1;
  for (int i = 0; i < AZ_TAPE_NFN; ++i)
    if (!strcmp(name, AZ_TAPE_FN_NAMES[i])) return i;
  return -1;
}

int az_tape_create(void** tape) {
  if (!tape) return AZ_ERR_ARG(61);
  *tape = new Tape();
  return AZ_OK;
}

int az_tape_destroy(void* tape) {
  delete (Tape*)tape;
  return AZ_OK;
}

int az_tape_add(void* tape, int kind, int fn, const void* words, int nwords) {
  Tape* t = (Tape*)tape;
  if (!t || nwords < 0 || nwords > MAXW || (nwords && !words)) return AZ_ERR_ARG(62);
  Op op{}; op.kind = kind; op.fn = fn; op.nw = nwords;
  if (nwords) memcpy(op.w, words, sizeof(long) * nwords);
  if (kind == OP_CALL) { if (fn < 0 || fn >= AZ_TAPE_NFN || nwords != AZ_TAPE_FN_NARGS[fn]) return AZ_ERR_ARG(63); }
  else if (kind == OP_EVENT_RECORD || kind == OP_STREAM_WAIT) { if (nwords != 2) return AZ_ERR_ARG(64); }
  else if (kind != OP_BREAK) return AZ_ERR_ARG(65);
  t->ops.push_back(op);
  return AZ_OK;
}

long az_tape_play(void* tape, long start) {
  Tape* t = (Tape*)tape;
  if (!t || start < 0) return -1;
  const long n = (long)t->ops.size();
  for (long i = start; i < n; ++i) {
    const Op& op = t->ops[i];
    int rc = 0;
    switch (op.kind) {
      case OP_CALL: rc = az_tape_dispatch(op.fn, op.w); break;
      case OP_EVENT_RECORD: { hipError_t e = hipEventRecord((hipEvent_t)op.w[0], (hipStream_t)op.w[1]); rc = e == hipSuccess ? 0 : -(int)e; break; }
      case OP_STREAM_WAIT: { hipError_t e = hipStreamWaitEvent((hipStream_t)op.w[0], (hipEvent_t)op.w[1], 0); rc = e == hipSuccess ? 0 : -(int)e; break; }
      case OP_BREAK: return i + 1;
    }
    if (rc) {
      // an entry point that failed between `set stop event` and `clear stop event` must not leave the event armed on this thread:
      // every later launch from it would re-record a stale fork event on whatever stream it runs on (result ignored on purpose:
      // 'the armed event was never carried' is exactly the state being cleaned up)
      (void)az_set_launch_stop_event(nullptr);
      t->err_index = i; t->err_rc = rc; return -2;
    }
  }
  return n;
}

int az_tape_last_error(void* tape, long* index, int* rc) {
  Tape* t = (Tape*)tape;
  if (!t) return AZ_ERR_ARG(61);
  if (index) *index = t->err_index;
  if (rc) *rc = t->err_rc;
  return AZ_OK;
}

}  // extern "C"
