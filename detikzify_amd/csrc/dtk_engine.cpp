// dtk_engine.cpp — the run loop of a batch of rollouts, native (include/dtk.h, "dtk_engine_*").
//
// Host code only: no HIP call in this file.  The device is reached through dtk_engine_ops — for dtk_engine_create these are the
// library's own entry points (dtk_decode_batch_launch / _wait, dtk_prefill_slot, dtk_kv_fork, dtk_resume_slot, ...), for the CPU
// tests a scripted device.  What it replaces is the per-token host iteration of HF GenerationMixin._sample under
// DetikzifyGenerator.generate / rollout (reference detikzify/infer/generate.py:246-282): with 64 rollouts in one batch that loop —
// 64 Python threads woken per token behind one interpreter lock — left the GPU idle a third of the time.
//
// One thread per engine owns the context's main stream:
//   * keeps `depth` (2) steps in flight: step k+1 is queued on the device before step k's tokens are read, so the device never
//     waits for the host between steps;
//   * appends each slot's token to that slot's ring and applies the sequence's own stop rules (stop ids, token budget);
//   * wakes a slot's reader only at a flush token (a newline), after flush_max tokens, or at the end of the sequence;
//   * executes queued joins (resume in place | fork the image prefix | prefill) between steps, after the steps in flight have been
//     collected — a prefill drains the stream anyway, and a resumed slot must not be part of a step in flight.
// A slot's arithmetic never depends on its company (csrc/dtk_api.hip), so which steps a sequence shares with which others —
// the only thing this loop decides differently from the Python engine it replaces — does not change a token.
#include <array>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/dtk.h"

namespace {

using Clock = std::chrono::steady_clock;
double now_s() { return std::chrono::duration<double>(Clock::now().time_since_epoch()).count(); }

struct Seq {
  int state = 0;            // 0 = no sequence yet, else DTK_SEQ_*
  bool decoding = false;    // takes part in further steps
  std::vector<int64_t> toks;
  size_t read = 0, flushed = 0;     // toks[0, read) delivered; toks[read, flushed) may be delivered
  int budget = 0, emitted = 0;      // tokens the sequence may emit / has emitted
  int need = 0, launched = 0;       // steps to launch for it in total (budget + the forced step of a resumed slot) / launched so far
  int skip = 0;                     // 1: the next collected token is the forced last prompt token of a resumed slot (dropped)
  int n_stop = 0;
  int64_t stop[8] = {};
  int flush_mode = 0, flush_max = 64;
  std::condition_variable cv;
};

struct Cmd {
  dtk_join* join = nullptr;
  int rc = 0;
  bool done = false;
  std::condition_variable cv;
};

using ActiveSet = std::array<int32_t, DTK_MAX_BATCH>;

}  // namespace

struct dtk_engine {
  dtk_engine_ops ops{};
  std::mutex mu;
  std::condition_variable cv_run;
  std::thread th;
  bool quit = false;
  std::deque<Cmd*> cmds;
  Seq seq[DTK_MAX_BATCH];
  std::deque<ActiveSet> inflight;
  int depth = 2;
  int gather_left = 0;
  Clock::time_point gather_deadline{};
  std::vector<uint8_t> flush_tok;
  dtk_engine_stats st{};
  int err_rc = 0;
  std::string err;
  double idle_from = -1.0;      // since when sequences are decoding with no step in flight (< 0: not the case)
};

namespace {

bool any_decoding(const dtk_engine* e) {
  for (const Seq& q : e->seq) if (q.decoding) return true;
  return false;
}

void mark_idle(dtk_engine* e) {
  if (e->inflight.empty() && e->idle_from < 0 && any_decoding(e)) e->idle_from = now_s();
}

// the device failed: nothing it holds can be trusted any more
void poison(dtk_engine* e, int rc) {
  if (e->err_rc) return;
  e->err_rc = rc ? rc : DTK_ERR_STATE;
  const char* t = e->ops.last_error ? e->ops.last_error(e->ops.dev) : nullptr;
  e->err = (t && *t) ? t : "device call failed";
  e->inflight.clear();
  for (Seq& q : e->seq) { q.decoding = false; q.cv.notify_all(); }
}

int launch_set(dtk_engine* e, ActiveSet& set) {
  int n = 0;
  for (int s = 0; s < DTK_MAX_BATCH; ++s) {
    Seq& q = e->seq[s];
    set[s] = (q.decoding && q.launched < q.need) ? 1 : 0;
    n += set[s];
  }
  return n;
}

void launch(dtk_engine* e, std::unique_lock<std::mutex>& lk, const ActiveSet& set, int n) {
  for (int s = 0; s < DTK_MAX_BATCH; ++s) if (set[s]) e->seq[s].launched++;
  const double t0 = now_s();
  if (e->inflight.empty() && e->idle_from >= 0) e->st.idle_s += t0 - e->idle_from;
  e->idle_from = -1.0;
  if (e->st.first_launch_t == 0.0) e->st.first_launch_t = t0;
  if (2 * n < e->ops.decode_slots) e->st.steps_below_half_occupancy++;
  e->inflight.push_back(set);
  lk.unlock();
  const int rc = e->ops.launch(e->ops.dev, set.data());
  const double t1 = now_s();
  lk.lock();
  e->st.launch_s += t1 - t0;
  if (rc) poison(e, rc);
}

void collect(dtk_engine* e, std::unique_lock<std::mutex>& lk, bool drain) {
  const ActiveSet set = e->inflight.front();
  int64_t toks[DTK_MAX_BATCH];
  lk.unlock();
  const double t0 = now_s();
  const int rc = e->ops.wait(e->ops.dev, toks);
  const double t1 = now_s();
  lk.lock();
  if (e->err_rc) return;             // (poisoned while we waited: the queue of steps is gone)
  (drain ? e->st.drain_s : e->st.wait_s) += t1 - t0;
  if (t1 - t0 < 1e-4) e->st.host_bound_steps++;      // the device had finished already: this step waited for the host
  e->inflight.pop_front();
  e->st.steps++;
  e->st.last_collect_t = t1;
  if (rc) { poison(e, rc); return; }
  for (int s = 0; s < DTK_MAX_BATCH; ++s) {
    if (!set[s]) continue;
    Seq& q = e->seq[s];
    if (!q.decoding) { e->st.wasted_slot_steps++; continue; }     // ended (stop id) or left while this step was in flight
    if (q.skip) { q.skip = 0; continue; }
    const int64_t tok = toks[s];
    q.toks.push_back(tok);
    q.emitted++;
    e->st.tokens_out++;
    bool end = q.emitted >= q.budget;
    for (int k = 0; k < q.n_stop; ++k) end = end || tok == q.stop[k];
    if (end) {
      q.decoding = false;
      q.state = DTK_SEQ_FINISHED;
      q.flushed = q.toks.size();
      q.cv.notify_all();
    } else if (q.flush_mode == 0 || (tok >= 0 && (size_t)tok < e->flush_tok.size() && e->flush_tok[(size_t)tok]) ||
               q.toks.size() - q.flushed >= (size_t)q.flush_max) {
      q.flushed = q.toks.size();
      q.cv.notify_all();
    }
  }
  mark_idle(e);
}

int set_error(dtk_engine* e, dtk_join* j, int rc, const char* what) {
  const char* t = e->ops.last_error ? e->ops.last_error(e->ops.dev) : nullptr;
  snprintf(j->error_out, sizeof j->error_out, "%s%s%s", what, (t && *t) ? ": " : "", (t && *t) ? t : "");
  return rc;
}

// the device side of a join; runs on the loop's thread with no step in flight and e->mu released
int do_join(dtk_engine* e, dtk_join* j) {
  const dtk_engine_ops& o = e->ops;
  int slot = j->slot;
  j->slot_out = slot;
  if (j->try_resume && j->n_ids >= 2) {
    int best = -1, best_len = 0;
    const int nc = j->n_candidates > 0 ? j->n_candidates : 1;
    for (int k = 0; k < nc; ++k) {
      const int cand = j->n_candidates > 0 ? j->candidates[k] : slot;
      if (cand < 0 || cand >= o.decode_slots) return set_error(e, j, DTK_ERR_ARG, "dtk_engine_join: candidate slot out of range");
      int lcp = 0;
      const int rc = o.slot_lcp(o.dev, cand, j->ids, j->n_ids, j->image_key, &lcp);
      if (rc) return set_error(e, j, rc, "dtk_slot_lcp");
      if (lcp > best_len) { best = cand; best_len = lcp; }
    }
    if (best >= 0 && best_len >= j->n_ids - 1) {
      j->slot_out = slot = best;
      int rc = o.set_sampling_slot(o.dev, slot, &j->sampling);
      if (rc) return set_error(e, j, rc, "dtk_set_sampling_slot");
      rc = o.resume_slot(o.dev, slot, j->ids, j->n_ids, j->image_key);
      if (rc) return set_error(e, j, rc, "dtk_resume_slot");
      j->how_out = DTK_JOIN_RESUMED;
      return DTK_OK;
    }
  }
  int rc = o.set_sampling_slot(o.dev, slot, &j->sampling);
  if (rc) return set_error(e, j, rc, "dtk_set_sampling_slot");
  const int reuse = DTK_PREFILL_REUSE_PREFIX | DTK_PREFILL_REUSE_IMAGE;
  if (j->prefix_len > 0 && j->prefix_in_place) {
    rc = o.prefill_slot(o.dev, slot, j->ids, j->n_ids, j->pixels, j->image_key, reuse);
    if (rc) return set_error(e, j, rc, "dtk_prefill_slot");
    j->how_out = DTK_JOIN_IN_PLACE;
    return DTK_OK;
  }
  if (j->prefix_len > 0 && j->prefix_src >= 0) {
    if (j->prefix_len > j->n_ids) return set_error(e, j, DTK_ERR_ARG, "dtk_engine_join: prefix longer than the prompt");
    if (j->prefix_encode) {     // the image's first rollout: ViT + prefix prefill into the prefix-cache slot, greedy (its logits are forked)
      dtk_sampling g{};
      g.temperature = 1.0f; g.top_p = 1.0f;
      rc = o.set_sampling_slot(o.dev, j->prefix_src, &g);
      if (rc) return set_error(e, j, rc, "dtk_set_sampling_slot(prefix slot)");
      rc = o.prefill_slot(o.dev, j->prefix_src, j->ids, j->prefix_len, j->pixels, j->image_key, 0);
      if (rc) return set_error(e, j, rc, "dtk_prefill_slot(prefix slot)");
    }
    rc = o.kv_fork(o.dev, j->prefix_src, slot, j->prefix_len);
    if (rc) return set_error(e, j, rc, "dtk_kv_fork");
    if (j->prefix_src_whole && j->n_ids == j->prefix_len) {
      j->how_out = DTK_JOIN_FORK_WHOLE;       // KV and next-token logits came with the fork
      return DTK_OK;
    }
    rc = o.prefill_slot(o.dev, slot, j->ids, j->n_ids, j->pixels, j->image_key, reuse);
    if (rc) return set_error(e, j, rc, "dtk_prefill_slot");
    j->how_out = DTK_JOIN_FORK_TAIL;
    return DTK_OK;
  }
  rc = o.prefill_slot(o.dev, slot, j->ids, j->n_ids, j->pixels, j->image_key, j->full_flags);
  if (rc) return set_error(e, j, rc, "dtk_prefill_slot");
  j->how_out = DTK_JOIN_FULL;
  return DTK_OK;
}

void exec_join(dtk_engine* e, std::unique_lock<std::mutex>& lk, Cmd* c) {
  dtk_join* j = c->join;
  int rc = DTK_OK;
  j->error_out[0] = 0;
  j->slot_out = j->slot;
  j->how_out = -1;
  auto busy = [&](int s) { return s < 0 || s >= e->ops.decode_slots || s >= DTK_MAX_BATCH || e->seq[s].decoding; };
  if (e->err_rc) {
    rc = e->err_rc;
    snprintf(j->error_out, sizeof j->error_out, "the engine's device failed earlier: %s", e->err.c_str());
  } else if (!j->ids || j->n_ids < 1 || j->n_stop < 0 || j->n_stop > 8 || j->n_candidates < 0 || j->n_candidates > DTK_MAX_BATCH) {
    rc = DTK_ERR_ARG;
    snprintf(j->error_out, sizeof j->error_out, "dtk_engine_join: bad argument");
  } else if (busy(j->slot)) {
    rc = DTK_ERR_STATE;
    snprintf(j->error_out, sizeof j->error_out, "dtk_engine_join: slot %d is not a free decoding slot (0..%d)", j->slot, e->ops.decode_slots - 1);
  } else {
    for (int k = 0; k < j->n_candidates && !rc; ++k)
      if (busy(j->candidates[k])) {
        rc = DTK_ERR_STATE;
        snprintf(j->error_out, sizeof j->error_out, "dtk_engine_join: candidate slot %d is not a free decoding slot", j->candidates[k]);
      }
  }
  if (!rc) {
    lk.unlock();
    const double t0 = now_s();
    rc = do_join(e, j);
    const double t1 = now_s();
    lk.lock();
    e->st.join_s += t1 - t0;
    if (rc == DTK_ERR_HIP) poison(e, rc);
  }
  if (!rc) {
    Seq& q = e->seq[j->slot_out];
    q.toks.clear();
    q.read = q.flushed = 0;
    q.emitted = q.launched = 0;
    q.skip = j->how_out == DTK_JOIN_RESUMED ? 1 : 0;
    q.budget = j->max_new_tokens;
    // never past the KV capacity: every launched step appends one position
    const int room = e->ops.max_positions - e->ops.context_len_slot(e->ops.dev, j->slot_out);
    q.need = q.budget + q.skip;
    if (q.need > room) { q.need = room > 0 ? room : 0; q.budget = q.need - q.skip; }
    q.n_stop = j->n_stop;
    for (int k = 0; k < j->n_stop; ++k) q.stop[k] = j->stop_ids[k];
    q.flush_mode = j->flush_mode;
    q.flush_max = j->flush_max > 0 ? j->flush_max : 64;
    q.decoding = q.budget > 0;
    q.state = q.decoding ? DTK_SEQ_RUNNING : DTK_SEQ_FINISHED;
    e->st.joins++;
    if (q.skip) e->st.resumed++;
    if (e->gather_left > 0) e->gather_left--;
    mark_idle(e);
  }
  c->rc = rc;
  c->done = true;
  c->cv.notify_all();       // (the waiter owns `c`; it cannot run before this thread releases e->mu)
}

void run(dtk_engine* e) {
  std::unique_lock<std::mutex> lk(e->mu);
  for (;;) {
    if (e->quit) {
      while (!e->inflight.empty() && !e->err_rc) collect(e, lk, true);
      break;
    }
    if (!e->cmds.empty()) {
      while (!e->inflight.empty() && !e->err_rc) collect(e, lk, true);
      while (!e->cmds.empty()) {
        Cmd* c = e->cmds.front();
        e->cmds.pop_front();
        exec_join(e, lk, c);
      }
      continue;
    }
    if (e->err_rc) { e->cv_run.wait(lk); continue; }
    if (e->gather_left > 0 && Clock::now() >= e->gather_deadline) e->gather_left = 0;
    const bool gathering = e->gather_left > 0;
    ActiveSet set;
    const int n = gathering ? 0 : launch_set(e, set);
    if (n > 0 && (int)e->inflight.size() < e->depth) { launch(e, lk, set, n); continue; }
    if (!e->inflight.empty()) { collect(e, lk, false); continue; }
    if (gathering) e->cv_run.wait_until(lk, e->gather_deadline);
    else e->cv_run.wait(lk);
  }
}

// ---- the library's own context as the device --------------------------------------------------------------------------------------
int ctx_launch(void* d, const int32_t* a) { return dtk_decode_batch_launch((dtk_ctx*)d, a); }
int ctx_wait(void* d, int64_t* t) { return dtk_decode_batch_wait((dtk_ctx*)d, t); }
int ctx_prefill(void* d, int s, const int64_t* ids, int T, const float* px, uint64_t key, int flags) { return dtk_prefill_slot((dtk_ctx*)d, s, ids, T, px, key, flags, nullptr); }
int ctx_sampling(void* d, int s, const dtk_sampling* sp) { return dtk_set_sampling_slot((dtk_ctx*)d, s, sp); }
int ctx_fork(void* d, int a, int b, int n) { return dtk_kv_fork((dtk_ctx*)d, a, b, n); }
int ctx_lcp(void* d, int s, const int64_t* ids, int n, uint64_t key, int* out) { return dtk_slot_lcp((dtk_ctx*)d, s, ids, n, key, out); }
int ctx_resume(void* d, int s, const int64_t* ids, int n, uint64_t key) { return dtk_resume_slot((dtk_ctx*)d, s, ids, n, key); }
int ctx_len(void* d, int s) { return dtk_context_len_slot((const dtk_ctx*)d, s); }
const char* ctx_err(void* d) { return dtk_last_error((const dtk_ctx*)d); }

}  // namespace

#pragma GCC visibility push(default)
extern "C" {

int dtk_engine_create_ops(const dtk_engine_ops* ops, dtk_engine** out) {
  if (!ops || !out || !ops->launch || !ops->wait || !ops->prefill_slot || !ops->set_sampling_slot || !ops->kv_fork || !ops->slot_lcp ||
      !ops->resume_slot || !ops->context_len_slot || ops->decode_slots < 1 || ops->decode_slots > DTK_MAX_BATCH || ops->max_positions < 1)
    return DTK_ERR_ARG;
  dtk_engine* e = new dtk_engine();
  e->ops = *ops;
  e->th = std::thread(run, e);
  *out = e;
  return DTK_OK;
}

int dtk_engine_create(dtk_ctx* ctx, dtk_engine** out) {
  if (!ctx || !out) return DTK_ERR_ARG;
  dtk_engine_ops o{};
  o.dev = ctx;
  o.launch = ctx_launch; o.wait = ctx_wait; o.prefill_slot = ctx_prefill; o.set_sampling_slot = ctx_sampling; o.kv_fork = ctx_fork;
  o.slot_lcp = ctx_lcp; o.resume_slot = ctx_resume; o.context_len_slot = ctx_len; o.last_error = ctx_err;
  o.max_positions = dtk_max_positions(ctx);
  o.decode_slots = dtk_max_decode_slots(ctx);
  return dtk_engine_create_ops(&o, out);
}

void dtk_engine_destroy(dtk_engine* e) {
  if (!e) return;
  {
    std::lock_guard<std::mutex> g(e->mu);
    e->quit = true;
    e->cv_run.notify_all();
  }
  if (e->th.joinable()) e->th.join();
  {
    std::unique_lock<std::mutex> lk(e->mu);
    while (!e->cmds.empty()) {        // joins that were queued behind the stop
      Cmd* c = e->cmds.front();
      e->cmds.pop_front();
      snprintf(c->join->error_out, sizeof c->join->error_out, "the engine was destroyed");
      c->rc = DTK_ERR_STATE;
      c->done = true;
      c->cv.notify_all();
    }
    for (Seq& q : e->seq) {
      q.decoding = false;
      if (q.state == DTK_SEQ_RUNNING) q.state = DTK_SEQ_LEFT;
      q.flushed = q.toks.size();
      q.cv.notify_all();
    }
  }
  // readers that are still inside dtk_engine_read wake up under e->mu; give them the lock once more before the memory goes
  { std::lock_guard<std::mutex> g(e->mu); }
  delete e;
}

const char* dtk_engine_last_error(const dtk_engine* e) { return e ? e->err.c_str() : ""; }

int dtk_engine_set_flush_tokens(dtk_engine* e, const int64_t* ids, int n) {
  if (!e || n < 0 || (n > 0 && !ids)) return DTK_ERR_ARG;
  int64_t hi = -1;
  for (int i = 0; i < n; ++i) { if (ids[i] < 0 || ids[i] > (1 << 24)) return DTK_ERR_ARG; if (ids[i] > hi) hi = ids[i]; }
  std::lock_guard<std::mutex> g(e->mu);
  e->flush_tok.assign((size_t)(hi + 1), 0);
  for (int i = 0; i < n; ++i) e->flush_tok[(size_t)ids[i]] = 1;
  return DTK_OK;
}

int dtk_engine_set_option(dtk_engine* e, const char* name, int value) {
  if (!e || !name) return DTK_ERR_ARG;
  std::lock_guard<std::mutex> g(e->mu);
  if (!strcmp(name, "depth")) {
    if (value < 1 || value > 2) return DTK_ERR_ARG;
    e->depth = value;
    return DTK_OK;
  }
  return DTK_ERR_ARG;
}

int dtk_engine_expect(dtk_engine* e, int n, int timeout_ms) {
  if (!e || n < 0 || timeout_ms < 0) return DTK_ERR_ARG;
  std::lock_guard<std::mutex> g(e->mu);
  e->gather_left = n < e->ops.decode_slots ? n : e->ops.decode_slots;
  e->gather_deadline = Clock::now() + std::chrono::milliseconds(timeout_ms);
  e->st.first_launch_t = e->st.last_collect_t = 0.0;
  e->cv_run.notify_all();
  return DTK_OK;
}

int dtk_engine_submit(dtk_engine* e, dtk_join* j, uint64_t* ticket_out) {
  if (!e || !j || !ticket_out) return DTK_ERR_ARG;
  std::lock_guard<std::mutex> g(e->mu);
  if (e->quit) { snprintf(j->error_out, sizeof j->error_out, "the engine is being destroyed"); return DTK_ERR_STATE; }
  Cmd* c = new Cmd();
  c->join = j;
  e->cmds.push_back(c);
  e->cv_run.notify_all();
  *ticket_out = (uint64_t)(uintptr_t)c;
  return DTK_OK;
}

int dtk_engine_await(dtk_engine* e, uint64_t ticket) {
  if (!e || !ticket) return DTK_ERR_ARG;
  Cmd* c = (Cmd*)(uintptr_t)ticket;
  int rc;
  {
    std::unique_lock<std::mutex> lk(e->mu);
    c->cv.wait(lk, [&] { return c->done; });
    rc = c->rc;
    e->cv_run.notify_all();       // a sequence may have become decodable
  }
  delete c;
  return rc;
}

int dtk_engine_join(dtk_engine* e, dtk_join* j) {
  uint64_t t = 0;
  const int rc = dtk_engine_submit(e, j, &t);
  return rc ? rc : dtk_engine_await(e, t);
}

int dtk_engine_read(dtk_engine* e, int slot, int64_t* out, int cap, int32_t* n_out, int32_t* state_out, int timeout_ms) {
  if (!e || slot < 0 || slot >= DTK_MAX_BATCH || !out || cap < 1 || !n_out || !state_out) return DTK_ERR_ARG;
  std::unique_lock<std::mutex> lk(e->mu);
  Seq& q = e->seq[slot];
  *n_out = 0;
  *state_out = q.state;
  if (q.state == 0) return DTK_ERR_STATE;
  auto ready = [&] { return q.read < q.flushed || q.state != DTK_SEQ_RUNNING || e->err_rc != 0; };
  if (timeout_ms < 0) q.cv.wait(lk, ready);
  else q.cv.wait_for(lk, std::chrono::milliseconds(timeout_ms), ready);
  size_t n = q.flushed - q.read;
  if (n > (size_t)cap) n = (size_t)cap;
  if (n) memcpy(out, q.toks.data() + q.read, n * sizeof(int64_t));
  q.read += n;
  *n_out = (int32_t)n;
  if (n) e->st.reader_wakeups++;
  *state_out = (q.state != DTK_SEQ_RUNNING && q.read < q.toks.size()) ? DTK_SEQ_RUNNING : q.state;
  if (e->err_rc && n == 0 && q.state == DTK_SEQ_RUNNING) return e->err_rc;
  return DTK_OK;
}

int dtk_engine_leave(dtk_engine* e, int slot) {
  if (!e || slot < 0 || slot >= DTK_MAX_BATCH) return DTK_ERR_ARG;
  std::lock_guard<std::mutex> g(e->mu);
  Seq& q = e->seq[slot];
  q.decoding = false;
  if (q.state == DTK_SEQ_RUNNING) q.state = DTK_SEQ_LEFT;
  q.flushed = q.toks.size();
  q.cv.notify_all();
  if (!any_decoding(e)) e->idle_from = -1.0;
  e->cv_run.notify_all();
  return DTK_OK;
}

int dtk_engine_get_stats(dtk_engine* e, dtk_engine_stats* out) {
  if (!e || !out) return DTK_ERR_ARG;
  std::lock_guard<std::mutex> g(e->mu);
  *out = e->st;
  return DTK_OK;
}

}  // extern "C"
#pragma GCC visibility pop
