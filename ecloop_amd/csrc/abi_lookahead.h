// abi_lookahead.h — look-ahead over a caller's small contiguous jobs (host side of ecl_hip_add_range; included by ecloop_hip.hip).
//
// Why.  The reference hands its workers jobs of MAX_JOB_SIZE = 2^21 keys from one mutex-guarded counter (main.c:16,418-431): 0.17 ms of
// work for this GPU, where a launch only reaches the kernel's rate from 2^28 keys up (one inversion per lane and group against an
// oversubscribed chip, see call_geometry).  Bound through the ABI unchanged, the reference ran at 7.4 Gkeys/s (59 % of the library's
// rate, profiles/r05_ref_binding.txt) and every caller had to raise a #define to get the rest.  The library now hides the small jobs:
// when the calls that share a filter form the reference's pattern - equal-sized jobs, each starting where the one before ended -
// a call that finds nothing prepared runs ONE large sweep from its start (up to 2^30 keys at the large-call geometry), keeps the
// sweep's hit records on the host, sorted by key offset, and answers the following calls from them without a launch.
//
// What is guaranteed.  A call is answered from a sweep only if all of its keys lie inside the sweep; it receives exactly the records
// whose key offset falls into its own range, re-based to its own start - the same set a launch of its own would report (same kernel,
// same filter, same flags; the order of records within a call was never specified: the device appends them with atomicAdd).  A sweep
// whose records do not fit (more than 2^20 hits: a dense filter) is dropped and the pattern is left alone until it changes; a caller
// whose density of hits is known from its earlier calls gets sweeps sized to about 2^16 records.  Anything else - a job that does not
// continue the pattern, a caller-set geometry, a filter changed by ecl_hip_bloom_insert - takes the plain launch, as before.
//
// How far ahead.  With the scan's end known (ecl_hip_set_scan_end: the reference's ctx->range_e, main.c:420) a sweep never passes the job
// that contains the end, so nothing is computed that will not be asked for, and the first sweep starts with the second job.  Without it
// the sweep is at most half of what the pattern has already consumed (rounded down to a power of two jobs): a scan that stops
// right after a sweep has then cost at most 1.5x its keys at the sweep rate - still ahead of 2^21-key launches - and a long scan
// reaches 2^30-key sweeps after 2^31 keys.
//
// Several contexts (the reference's -t N: one worker thread per context, all pulling from the same counter).  Contexts opened with the
// same flags and stride whose filters have the same fingerprint (over all of their words, computed on the device) form a group; sweeps, pattern and records belong to the group, guarded
// by its mutex.  A job handed to one context is answered from a sweep another context ran; a context whose job lies in a sweep that is
// still running on ANOTHER device first claims the next stretch for its own GPU, so N GPUs run N consecutive sweeps at once (contexts
// on the same device wait instead: a second sweep there would only share the chip).
#pragma once

#define LA_REC_CAP (1u << 17)      /* records copied with the sweep itself; up to 2^20 are read from the device afterwards */
#define LA_TARGET_HITS 65536.0     /* expected records per sweep when the caller's hit density is known */

struct la_region {
  u256 start;
  u64 nkeys = 0;
  int dev = 0;
  bool ready = false;
  std::vector<ecl_found> recs;  // sorted by key_offset once ready; never changed afterwards
  u64 consumed = 0;
};

struct la_key {
  u32 flags, offs;
  u64 nwords, bloom_fp, list_n, list_fp;
  bool operator<(const la_key& o) const {
    if (flags != o.flags) return flags < o.flags;
    if (offs != o.offs) return offs < o.offs;
    if (nwords != o.nwords) return nwords < o.nwords;
    if (bloom_fp != o.bloom_fp) return bloom_fp < o.bloom_fp;
    if (list_n != o.list_n) return list_n < o.list_n;
    return list_fp < o.list_fp;
  }
};

struct la_group {
  std::mutex mu;
  std::condition_variable cv;
  int members = 0;
  std::deque<std::shared_ptr<la_region>> regions;
  // the pattern of the calls that found nothing prepared
  bool pat = false, blocked = false;
  u64 job_n = 0, streak_keys = 0;
  u256 next;  // where the next job of the pattern is expected to start
  double hits = 0, keys = 0;
  // several contexts on ONE device (the reference's -t N on a box with fewer GPUs): the first of them that swept keeps sweeping there -
  // one set of sweep-sized walk buffers per GPU (19 GB at 2^30 keys) instead of one per context; the others launch their own job when
  // it is not covered yet
  std::map<int, const void*> sweeper;
};

static std::mutex la_registry_mu;
static std::map<la_key, std::weak_ptr<la_group>> la_registry;

static u64 la_default_max() {
  const char* e = getenv("ECL_HIP_LOOKAHEAD_LOG2");  // 0 = off; default 30
  long l = 30;
  if (e && *e) l = atol(e);
  if (l <= 0) return 0;
  if (l < 22) l = 22;
  if (l > 32) l = 32;
  return 1ull << l;
}

// (the fingerprints of a context's filter and list - la_bloom_fp, la_list_fp - are computed where the words are resident: ecloop_hip.hip)
static void la_leave(ecl_hip* h) {
  if (!h->grp) return;
  {
    std::lock_guard<std::mutex> lk(h->grp->mu);
    --h->grp->members;
    auto it = h->grp->sweeper.find(h->dev);
    if (it != h->grp->sweeper.end() && it->second == h) h->grp->sweeper.erase(it);
  }
  h->grp.reset();
}
static void la_join(ecl_hip* h) {
  const la_key key = {h->flags, h->offs, h->bloom_words, h->la_bloom_fp, h->list_n, h->la_list_fp};
  std::lock_guard<std::mutex> lk(la_registry_mu);
  for (auto it = la_registry.begin(); it != la_registry.end();) it = it->second.expired() ? la_registry.erase(it) : std::next(it);
  std::shared_ptr<la_group> g = la_registry[key].lock();
  if (!g) la_registry[key] = g = std::make_shared<la_group>();
  {
    std::lock_guard<std::mutex> lg(g->mu);
    ++g->members;
  }
  h->grp = g;
}

// number of keys from scalar `from` to scalar `to` in units of the stride 2^offs, if that is a count below 2^62
static bool la_offset(const ecl_hip* h, const u256& from, const u256& to, u64* off) {
  u256 d = sc_add(to, sc_neg(from));
  const u32 o = h->offs;
  if (o <= 190) {  // j * 2^offs < 2^252 < n for j < 2^62: the difference is the shifted count itself
    if (o) {
      u256 low = d;
      for (u32 i = o; i < 256; ++i) low.w[i >> 6] &= ~(1ull << (i & 63));
      if (low.w[0] | low.w[1] | low.w[2] | low.w[3]) return false;
      u256 r = {{0, 0, 0, 0}};
      for (u32 i = o; i < 256; ++i)
        if ((d.w[i >> 6] >> (i & 63)) & 1) r.w[(i - o) >> 6] |= 1ull << ((i - o) & 63);
      d = r;
    }
  } else {
    for (u32 i = 0; i < o; ++i) d = sc_half(d);
  }
  if (d.w[1] | d.w[2] | d.w[3] | (d.w[0] >> 62)) return false;
  *off = d.w[0];
  return true;
}
static u256 la_advance(const ecl_hip* h, const u256& from, u64 keys) { return sc_add(from, sc_mul_u64(sc_pow2(h->offs), keys)); }

static u64 pow2floor(u64 v) {
  u64 r = 1;
  while (r * 2 <= v) r *= 2;
  return v ? r : 0;
}

// keys from scalar `at` to the nearest sweep held at or after it (0: `at` lies inside one): a new sweep must not run into it
static u64 la_room(const ecl_hip* h, const la_group& g, const u256& at) {
  u64 room = ~0ull, o;
  for (auto& c : g.regions) {
    if (la_offset(h, at, c->start, &o)) room = o < room ? o : room;
    else if (la_offset(h, c->start, at, &o) && o < c->nkeys) return 0;
  }
  return room;
}

// keys the next sweep should cover when it starts at scalar `at` (a multiple of the job size n; 0: none)
static u64 la_plan(const ecl_hip* h, const la_group& g, const u256& at, u64 n) {
  u64 lim = h->la_max / n;  // in jobs
  // a sweep has to replace many launches to be worth its latency - and must never turn "one job per GPU" (N equal contiguous shards, one
  // per context: what a host that knows its GPUs hands out) into one GPU sweeping all of them while the others wait
  const u64 least = 4ull * (u64)(g.members > 1 ? g.members : 1);
  const u64 room = la_room(h, g, at) / n;
  if (room < lim) lim = room;
  if (h->la_have_end) {
    // keys from `at` to the end as the reference counts them: plain integers (fe_cmp, main.c:420), a last partial stride rounded up
    if (u256_cmp(at, h->la_end) >= 0) return 0;  // the end lies behind: not the scan this hint was given for
    u256 d;
    u256_sub(d, h->la_end, at);
    const u32 o = h->offs;
    bool partial = false;
    u256 q = {{0, 0, 0, 0}};
    for (u32 i = 0; i < 256; ++i)
      if ((d.w[i >> 6] >> (i & 63)) & 1) {
        if (i < o) partial = true;
        else q.w[(i - o) >> 6] |= 1ull << ((i - o) & 63);
      }
    if (!(q.w[1] | q.w[2] | q.w[3] | (q.w[0] >> 62))) {
      const u64 e = q.w[0] + (partial ? 1 : 0);
      const u64 jobs = (e + n - 1) / n;  // the jobs that start before the end (main.c:420: a worker stops at range_s >= range_e)
      if (jobs < lim) lim = jobs;
    }
  } else {
    const u64 grown = pow2floor(g.streak_keys / n / 2);
    if (grown < lim) lim = grown;
  }
  if (g.keys > 0 && g.hits > 0) {
    const double jobs = LA_TARGET_HITS / (g.hits / g.keys) / (double)n;
    if (jobs < (double)lim) lim = (u64)jobs;
  }
  return lim < least ? 0 : lim * n;
}

static bool la_rec_less(const ecl_found& a, const ecl_found& b) {
  if (a.key_offset != b.key_offset) return a.key_offset < b.key_offset;
  if (a.compressed != b.compressed) return a.compressed > b.compressed;
  return a.endo < b.endo;
}

// runs the sweep [r->start, + r->nkeys) on h's GPU and fills r->recs; the group's lock is NOT held
static int la_sweep(ecl_hip* h, la_region* r) {
  if (h->la_buf.size() < LA_REC_CAP) h->la_buf.resize(LA_REC_CAP);
  u32 cnt = 0;
  int rc = add_core(h, r->start, r->nkeys, h->la_buf.data(), LA_REC_CAP, &cnt);
  if (rc == ECL_E_OVERFLOW) {
    if (cnt > h->last_held) return ECL_E_OVERFLOW;  // more hits than the device kept: this filter is too dense to look ahead
    r->recs.resize(cnt);
    memcpy(r->recs.data(), h->la_buf.data(), (size_t)LA_REC_CAP * sizeof(ecl_found));
    u32 got = 0;
    rc = ecl_hip_fetch_found(h, LA_REC_CAP, r->recs.data() + LA_REC_CAP, cnt - LA_REC_CAP, &got);
    if (rc != ECL_OK) return rc;
    if (got != cnt - LA_REC_CAP) return ECL_E_OVERFLOW;
  } else if (rc == ECL_OK) {
    r->recs.assign(h->la_buf.begin(), h->la_buf.begin() + cnt);
  } else {
    return rc;
  }
  std::sort(r->recs.begin(), r->recs.end(), la_rec_less);
  h->la_sweeps += 1, h->la_swept_keys += r->nkeys;
  return ECL_OK;
}

static bool la_may_sweep(const ecl_hip* h, const la_group& g) {
  const auto it = g.sweeper.find(h->dev);
  return it == g.sweeper.end() || it->second == h;
}

// claims [at, at + L) for this context, runs it, publishes it.  Called and left with the group's lock held.
static bool la_claim_and_sweep(ecl_hip* h, la_group& g, std::unique_lock<std::mutex>& lk, const u256& at, u64 L) {
  g.sweeper[h->dev] = h;
  auto r = std::make_shared<la_region>();
  r->start = at, r->nkeys = L, r->dev = h->dev;
  g.regions.push_back(r);
  g.next = la_advance(h, at, L);
  lk.unlock();
  const int rc = la_sweep(h, r.get());
  lk.lock();
  if (rc == ECL_OK) {
    r->ready = true;
    g.hits += (double)r->recs.size(), g.keys += (double)L;
    while (g.regions.size() > (size_t)(2 * g.members + 2) && g.regions.front()->ready) g.regions.pop_front();  // stretches nobody came back for
  } else {  // too dense, a scan through the scalar 0, out of memory ...: this pattern goes on with plain launches
    for (auto it = g.regions.begin(); it != g.regions.end(); ++it)
      if (it->get() == r.get()) {
        g.regions.erase(it);
        break;
      }
    g.blocked = true, g.next = at;
    (void)hipGetLastError();
  }
  g.cv.notify_all();
  return rc == ECL_OK;
}

static int la_fetch(ecl_hip* h, uint32_t first, ecl_found* out, uint32_t n, uint32_t* got) {
  if (!h->last_region || first >= h->last_host_n || n == 0) return ECL_OK;
  const u32 take = h->last_host_n - first < n ? h->last_host_n - first : n;
  const ecl_found* src = h->last_region->recs.data() + h->last_host_at + first;
  for (u32 i = 0; i < take; ++i) out[i] = src[i], out[i].key_offset -= h->last_host_off;
  *got = take;
  return ECL_OK;
}

// ecl_hip_add_range comes through here first.  *served = the call has been answered (rc is its result); otherwise the caller
// launches the job itself.
static int la_add_range(ecl_hip* h, const u256& k0, u64 n, ecl_found* out, u32 cap, u32* nout, bool* served) {
  *served = false;
  if (!h->la_max || h->geom_fixed || !h->la_key_valid || n > h->la_max / 4) return ECL_OK;
  if (!h->grp) la_join(h);
  la_group& g = *h->grp;
  std::unique_lock<std::mutex> lk(g.mu);
  bool counted = false;
  for (;;) {
    std::shared_ptr<la_region> r;
    u64 off = 0;
    for (auto& c : g.regions) {
      u64 o;
      if (la_offset(h, c->start, k0, &o) && o < c->nkeys && n <= c->nkeys - o) {
        r = c, off = o;
        break;
      }
    }
    if (r && r->ready) {
      ecl_found lo, hi;
      lo.key_offset = off, hi.key_offset = off + n;
      const auto cmp = [](const ecl_found& a, const ecl_found& b) { return a.key_offset < b.key_offset; };
      const size_t i0 = std::lower_bound(r->recs.begin(), r->recs.end(), lo, cmp) - r->recs.begin();
      const size_t i1 = std::lower_bound(r->recs.begin(), r->recs.end(), hi, cmp) - r->recs.begin();
      const u32 cnt = (u32)(i1 - i0), take = cnt < cap ? cnt : cap;
      for (u32 i = 0; i < take; ++i) out[i] = r->recs[i0 + i], out[i].key_offset -= off;
      h->last_region = r, h->last_host_at = i0, h->last_host_n = cnt, h->last_host_off = off, h->last_from_host = true;
      h->last_held = h->last_total = 0;
      r->consumed += n;
      if (r->consumed >= r->nkeys)
        for (auto it = g.regions.begin(); it != g.regions.end(); ++it)
          if (it->get() == r.get()) {
            g.regions.erase(it);
            break;
          }
      if (!counted) g.streak_keys += n;
      h->la_served_calls += 1, h->la_served_keys += n;
      *nout = cnt, *served = true;
      return cnt > cap ? ECL_E_OVERFLOW : ECL_OK;
    }
    if (r) {  // inside a sweep that is still running
      bool mine_busy = false;
      for (auto& c : g.regions) mine_busy |= !c->ready && c->dev == h->dev;
      // (... as long as the group does not already hold a sweep per context and one more: a device that runs far ahead of the others would
      // only push stretches nobody has come for yet out of the list)
      const bool crowded = g.regions.size() > (size_t)(g.members > 1 ? g.members : 1);
      const u64 L = (!mine_busy && !crowded && !g.blocked && g.pat && n == g.job_n && la_may_sweep(h, g)) ? la_plan(h, g, g.next, n) : 0;
      if (L) {
        const u256 at = g.next;
        (void)la_claim_and_sweep(h, g, lk, at, L);
      } else {
        g.cv.wait(lk);
      }
      continue;
    }
    // nothing prepared for this job
    if (counted) return ECL_OK;  // (a sweep of ours failed, or did not cover the job after all)
    counted = true;
    u64 gap = 0;
    // one context: the next job starts exactly where the last one ended; several worker threads: their calls arrive a few jobs out of order
    const u64 tol = g.members > 1 ? 4ull * (u64)g.members * n : 0;
    const bool ahead = g.pat && n == g.job_n && la_offset(h, g.next, k0, &gap) && gap <= tol;
    // ... or late: a job from before the front that no sweep covers (its worker was slow to call, or it fell between two sweeps)
    u64 span = 0, o;
    for (auto& c : g.regions)
      if (la_offset(h, c->start, g.next, &o) && o > span) span = o;
    const bool late = !ahead && g.pat && n == g.job_n && la_offset(h, k0, g.next, &gap) && gap <= tol + span;
    if (!ahead && !late) {  // a new pattern starts with this job
      g.pat = true, g.blocked = false, g.job_n = n, g.streak_keys = n, g.hits = g.keys = 0;
      g.next = la_advance(h, k0, n);
      return ECL_OK;
    }
    g.streak_keys += n;
    if (late) return ECL_OK;  // launched on its own
    const u64 L = g.blocked || !la_may_sweep(h, g) ? 0 : la_plan(h, g, k0, n);
    if (!L) {
      g.next = la_advance(h, k0, n);
      return ECL_OK;
    }
    if (!la_claim_and_sweep(h, g, lk, k0, L)) {
      g.next = la_advance(h, k0, n);
      return ECL_OK;
    }
  }
}

static void la_note_plain_call(ecl_hip* h, u64 n, u32 hits) {
  if (!h->grp) return;
  std::lock_guard<std::mutex> lk(h->grp->mu);
  h->grp->hits += hits, h->grp->keys += (double)n;
}

// ecl_hip_add_range after its argument checks: answered from a sweep, or launched as given
static int la_dispatch(ecl_hip* h, const u256& k0, u64 nkeys, ecl_found* out, u32 cap, u32* nout) {
  bool served = false;
  int rc = la_add_range(h, k0, nkeys, out, cap, nout, &served);  // a job inside a sweep this context (or one that shares its filter) has run
  if (served || rc != ECL_OK) return rc;
  rc = add_core(h, k0, nkeys, out, cap, nout);
  la_note_plain_call(h, nkeys, *nout);
  return rc;
}

extern "C" int ecl_hip_set_lookahead(ecl_hip* h, uint64_t max_keys) {
  if (!h || (max_keys && (max_keys < (1ull << 22) || max_keys > (1ull << 32)))) return ECL_E_ARG;
  h->la_max = max_keys;
  if (!max_keys) la_leave(h);
  return ECL_OK;
}
extern "C" int ecl_hip_set_scan_end(ecl_hip* h, const uint64_t end[4]) {
  if (!h) return ECL_E_ARG;
  h->la_have_end = end != nullptr;
  if (end) h->la_end = sc_reduce(u256_from(end));
  return ECL_OK;
}
extern "C" int ecl_hip_get_lookahead_stats(ecl_hip* h, uint64_t* sweeps, uint64_t* swept_keys, uint64_t* served_calls, uint64_t* served_keys) {
  if (!h) return ECL_E_ARG;
  if (sweeps) *sweeps = h->la_sweeps;
  if (swept_keys) *swept_keys = h->la_swept_keys;
  if (served_calls) *served_calls = h->la_served_calls;
  if (served_keys) *served_keys = h->la_served_keys;
  return ECL_OK;
}
