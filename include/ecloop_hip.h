/*
 * ecloop_hip.h — C ABI of the MI355X (gfx950) key-search engine: the drop-in boundary for ecloop's hot path.
 *
 * The reference (vladkens/ecloop v0.5.0) has no plugin/FFI seam: the hot path is reached by static calls inside
 * one translation unit.  This header cuts the seam at the L4 -> L3 call:
 *
 *   reference call site                                   replaced by
 *   ---------------------------------------------------   ------------------------------------------
 *   batch_add(ctx, pk, iterations)          main.c:349    ecl_hip_add_range()
 *     + check_found_add / check_hash        main.c:278-347  (device: hash160 + bloom probe; hits returned)
 *   ctx_precompute_gpoints(ctx)             main.c:219    done inside ecl_hip_open()/first add_range
 *   ec_gtable_init, ec_gtable_mul xN        ecc.c:876-929
 *     + ec_jacobi_grprdc + check_found_mul  main.c:531-534  ecl_hip_mul_batch()  (window tables built on the device)
 *   SHA-256 of each line under -raw         main.c:505-527  ecl_hip_mul_batch_raw()  (hashed on the device)
 *   qsort + blf_add loop of load_filter     main.c:112-131  optional: ecl_hip_sort_list() + ecl_hip_bloom_insert()
 *   blf_has(&ctx->blf, h)                   utils.c:308   device probe of the bits given to ecl_hip_set_bloom()
 *   bsearch(to_find_hashes) in ctx_check_hash main.c:212-216  optional: ecl_hip_set_list() (else on the host, as before)
 *   ctx->check_addr33/65, use_endo, ord_offs main.c:32-34,67  flags / ord_offs of ecl_hip_open()
 *
 * What stays on the host, unchanged in meaning: filter loading (main.c:71-131), the sorted-list confirm after a
 * bloom hit (main.c:212-216), calc_priv (main.c:267-276), pk_verify_hash (main.c:248-263), the found sink and
 * status line (main.c:134-203), the job scheduler (main.c:405-454).  The device reports bloom hits as
 * {key offset, endo index, address type, hash160}; the host turns them into private keys.
 *
 * Plain C types only: pointers, sizes, fixed-width integers.  No stdout/stderr/exit inside the library.
 *
 * Threads.  One handle = one context on one GPU; a handle must be used by ONE host thread at a time (any thread, not two at once);
 * different handles - on the same GPU or on different ones - may be used concurrently without any locking by the caller: one thread
 * per context is the intended pattern (the keyspace is range-partitioned, there is no collective).  What contexts of a process share
 * is guarded inside the library: the window table of `mul` (one per device and width: built once under a mutex by the first context
 * that needs it, read-only from then on, reference-counted, freed with the last context that holds it; a context that switches width
 * waits for its own kernels on both of its streams first), the once-per-process self-test record, and the look-ahead groups of
 * section 3 (sweeps and their records belong to the group of contexts that share a filter; one mutex per group; records are immutable
 * once published).  ecl_hip_strerror returns static text; ecl_hip_last_error returns the handle's own buffer (same thread rule).
 *
 * Exports: exactly the ecl_hip_* functions declared below (the library is linked with a version script; `nm -D` shows nothing else).
 */
#ifndef ECLOOP_HIP_H
#define ECLOOP_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: exactly the functions declared here are exported */
#pragma GCC visibility push(default)

typedef struct ecl_hip ecl_hip; /* opaque per-device context */

/* flags of ecl_hip_open: which encodings to hash (main.c:819-827) and the endomorphism switch (main.c:829) */
#define ECL_ADDR33 1u
#define ECL_ADDR65 2u
#define ECL_ENDO 4u

/* return codes */
#define ECL_OK 0
#define ECL_E_ARG (-1)      /* bad argument */
#define ECL_E_HIP (-2)      /* HIP runtime error: ecl_hip_last_error() has the text */
#define ECL_E_NODEV (-3)    /* no such device / no gfx950 code object for it */
#define ECL_E_OVERFLOW (-4) /* more hits than `cap`: *nout = total hits, only `cap` records were written (the rest: ecl_hip_fetch_found) */
#define ECL_E_NOBLOOM (-5)  /* add/mul called before ecl_hip_set_bloom */
#define ECL_E_RANGE (-6)    /* range touches the scalar 0 (mod n) neighbourhood the method cannot represent */
#define ECL_E_SELFTEST (-7) /* the device code failed its known-answer / cross-path self-test (miscompile, bad GPU) */

/* One bloom-filter hit.  key_offset counts keys from the `start` scalar of the call in units of the stride:
   privkey = start + key_offset * 2^ord_offs (mod n), then the endo map of calc_priv (main.c:267-276):
   endo 0: k, 1: -k, 2: k*lambda, 3: -k*lambda, 4: k*lambda^2, 5: -k*lambda^2.
   For ecl_hip_mul_batch key_offset is the index into the scalar array and endo is 0.
   h160 uses the reference's h160_t convention (addr.c:16): word i = digest bytes 4i..4i+3, big-endian. */
typedef struct ecl_found {
  uint64_t key_offset;
  uint32_t h160[5];
  uint8_t endo;
  uint8_t compressed; /* 1 = addr33, 0 = addr65 */
  uint8_t pad[2];
} ecl_found; /* 32 bytes */

/* ==== 1. THE SEAM: what a binding of the reference's host program calls (INTEGRATION.md shows it: six one-line edits of main.c) ====
   device_count / open / set_bloom / add_range / mul_batch / close replace the reference's calls listed above; fetch_found serves a call
   whose hits did not fit the caller's buffer; strerror / last_error give the text of a failure.  Everything after this section is
   optional: a caller that uses nothing else gets the full search path, at the library's rate (the look-ahead of section 3 is on by
   default). */
int ecl_hip_device_count(void);

/* Create a context on `device`. ord_offs: stride between consecutive keys is 2^ord_offs (0..255, main.c:221-222).
   On ECL_E_ARG / ECL_E_NODEV *out is untouched.  On any later failure (ECL_E_HIP, ECL_E_SELFTEST) *out holds a context
   that can only be asked for ecl_hip_last_error() and must be given to ecl_hip_close(). */
int ecl_hip_open(ecl_hip **out, int device, uint32_t flags, uint32_t ord_offs);
void ecl_hip_close(ecl_hip *h);

/* Copy the bloom bit array (little-endian u64 words, exactly the payload of a .blf file / of the in-memory
   filter built from a hash list, utils.c:277-280) into HBM.  May be called again to replace the filter. */
int ecl_hip_set_bloom(ecl_hip *h, const uint64_t *bits, uint64_t nwords);

/* Hash the nkeys keys  start, start+s, ..., start+(nkeys-1)*s  (s = 2^ord_offs; every encoding / endo variant
   selected at open) and report every bloom hit.  start: 256-bit scalar, 4 little-endian u64 limbs, as `fe`.
   Exactly these keys are tested - the caller reproduces the reference's job rounding (main.c:442,368).
   Consecutive calls whose `start` continues the previous range reuse the on-device walk state; small ones are answered from a
   look-ahead sweep (ecl_hip_set_lookahead) - same records either way.
   Returns ECL_OK, ECL_E_OVERFLOW (see above) or an error; ECL_E_ARG for an nkeys the geometry cannot walk in one call
   (more than 2^63, or more than 2^32 groups per lane - only reachable with a tiny caller-set geometry). */
int ecl_hip_add_range(ecl_hip *h, const uint64_t start[4], uint64_t nkeys, ecl_found *out, uint32_t cap,
                      uint32_t *nout);

/* `mul` command body (main.c:530-534): public keys of n scalars, hash, probe; key_offset = scalar index.
   The fixed-base window table of ec_gtable_mul (lib/ecc.c:876-929) is built on the device at a window width sized for
   HBM rather than for a CPU cache (the reference: 14 bits, 19.9 MB), with signed digits (a row holds 2^(W-1) points, a digit above
   2^(W-1) adds the negated point and carries): a context starts on 22 bits (12 rows, 11 * 2^21 + 2^14 points of 64 bytes = 1.5 GB,
   ~40 ms with the first call) and moves to 26 bits (10 rows, 19.6 GB, ~90 ms: 10 additions per scalar instead of 19) once it has
   multiplied 2^30 scalars, which is when the wider table has paid for its build (any width 8...29 can be fixed with
   ecl_hip_set_mul_window; 29 bits = 9 additions per scalar from a 138 GB table, for runs of 10^11 scalars and more).  A table is checked against the double-and-add kernel
   on sample slots before use (ECL_E_SELFTEST on a mismatch), shared between the contexts of a device in the process
   and freed with the last of them.  Results do not depend on the width. */
int ecl_hip_mul_batch(ecl_hip *h, const uint64_t (*scalars)[4], uint32_t n, ecl_found *out, uint32_t cap,
                      uint32_t *nout);
/* The records of the LAST ecl_hip_add_range / ecl_hip_mul_batch(_raw) call of this context that did not fit its `out`: a call
   keeps up to max(cap, 2^20) records on the device (32 MB), so after ECL_E_OVERFLOW (*nout = total > cap) the caller reads
   records [first, first + n) - the call itself delivered [0, cap) - instead of repeating a launch that may have walked
   2^32 keys.  *got = records written to `out` (fewer than n only if the call produced more than the device kept, i.e. more
   than max(cap, 2^20): then - and only then - the call has to be repeated with cap = total, as before).  Valid until the
   next add / mul call on the context.  The reference has no counterpart: its sink writes every hit as it is found
   (ctx_write_found, main.c:182-203), there is no buffer to overflow. */
int ecl_hip_fetch_found(ecl_hip *h, uint32_t first, ecl_found *out, uint32_t n, uint32_t *got);

const char *ecl_hip_strerror(int code);
const char *ecl_hip_last_error(const ecl_hip *h);

/* ==== 2. OPTIONAL: host-side steps of the path that can run on the device as well ====
   the exact list confirm (main.c:212-216), pk_verify_hash for a batch (main.c:248-263), `mul -raw`'s SHA-256 of the lines
   (main.c:505-527), load_filter's sort + bit setting for long lists (main.c:112-131), blf-gen's insert loop (utils.c:455-470);
   page-locked host memory for callers that feed `mul` */
/* Optional exact confirm on the device: the second half of ctx_check_hash (main.c:212-216).  h160 = the n sorted,
   unique list entries (ctx->to_find_hashes, order of compare_160, addr.c:18-26).  With a list resident, add_range /
   mul_batch report a hash only if it passed the bloom probe AND is in the list (a small kernel after the search
   kernel looks the bloom hits up by binary search; the list takes 20 bytes per entry of HBM); n = 0 removes the
   list again (bloom-only reporting).  ECL_E_ARG if the entries are not strictly increasing.  In list mode one call
   can stage max(cap, 2^20) bloom hits; beyond that it returns ECL_E_OVERFLOW with *nout = the number of bloom hits
   (and nothing to fetch: repeat the call with cap = *nout). */
int ecl_hip_set_list(ecl_hip *h, const uint32_t (*h160)[5], uint64_t n);

/* pk_verify_hash (main.c:248-263) for n reported private keys in one go: both hash160 values of k*G, derived on the
   device by a path other than the walk kernel (fixed-base window sum + own inversion per key).  The window sum and its
   table also give the base centre of a non-contiguous walk, and a hit shares its high digits with that centre, so
   the context self-test checks the window sum against the double-and-add kernel on full-width scalars.  ok[i] = 0
   for k = 0 (mod n).  The caller compares with the hit's h160 and treats a mismatch as fatal, like the reference. */
int ecl_hip_verify(ecl_hip *h, const uint64_t (*k)[4], uint32_t n, uint32_t (*h33)[5], uint32_t (*h65)[5], uint8_t *ok);

/* `mul -raw` (main.c:505-527: the scalar of a line is the SHA-256 of its bytes, read as a big-endian number): the same as
   ecl_hip_mul_batch with the hashing done on the device.  `text` holds the lines' bytes (anywhere, in any order, newline
   bytes or not), lines[i] = offset of line i in `text` (low 32 bits) | its length in bytes (high 32 bits); n <= 2^26 lines
   and text_bytes < 2^32 - 16 per call, every line inside the text (ECL_E_ARG otherwise).  key_offset of a hit = line
   index; the caller re-derives that line's private key (one SHA-256) for the found record.  Page-locked `text` / `lines`
   arrays are read by DMA directly; a call is pipelined like ecl_hip_mul_batch's (the table and the hashing piece by piece), so large
   calls (2^24 lines) run at the rate of large ecl_hip_mul_batch calls. */
int ecl_hip_mul_batch_raw(ecl_hip *h, const uint8_t *text, uint32_t text_bytes, const uint64_t *lines, uint32_t n, ecl_found *out,
                          uint32_t cap, uint32_t *nout);

/* load_filter's list preparation (main.c:112-124: qsort by compare_160, then duplicates removed - the same here) on the device:
   sorts the n entries of h160 in place into compare_160 order (addr.c:18-26: word by word), removes duplicates, *kept =
   entries left at the front of the array.  (Five stable 32-bit radix sorts of an index permutation, least significant word first,
   then adjacent-duplicate flags, a scan and a scatter: this library's own kernels, off the hot path - a list is prepared once per
   run.)  n < 2^31.  10^7 entries: 13 s of qsort + bit setting on the host, 0.05 s here (the bits: ecl_hip_set_bloom of a zero filter + ecl_hip_bloom_insert + ecl_hip_get_bloom). */
int ecl_hip_sort_list(ecl_hip *h, uint32_t (*h160)[5], uint64_t n, uint64_t *kept);

/* blf_add (utils.c:290-306) in bulk: set the 20 bits of each of n hash160 values (h160_t words) in the resident
   filter; ecl_hip_get_bloom copies the bit array back (e.g. to write a .blf file, utils.c:328-360). */
int ecl_hip_bloom_insert(ecl_hip *h, const uint32_t (*h160)[5], uint64_t n);
int ecl_hip_get_bloom(ecl_hip *h, uint64_t *bits, uint64_t nwords);
/* The insert loop of blf-gen (utils.c:455-470) for n hashes in input order: a hash that blf_has already reports at its
   turn is skipped; *added = the number that were new - the reference's "added N new items", exact also for duplicate
   and colliding hashes (the bits are those of ecl_hip_bloom_insert). */
int ecl_hip_bloom_insert_count(ecl_hip *h, const uint32_t (*h160)[5], uint64_t n, uint64_t *added);

/* Page-locked host memory (hipHostMalloc / hipHostFree): scalar arrays given to ecl_hip_mul_batch from such memory are read by the
   GPU's copy engine directly (57 GB/s), pageable ones go through a staging copy (18 GB/s: 0.57 instead of 1.3 G scalars/s for `mul`).
   The runtime places the pages next to the GPU, which matters on a two-socket host: from the far socket the copy runs at about half
   the rate (30 against 57 GB/s measured).  (There is no call that page-locks the caller's own memory in place: register / unregister
   cycles on memory the host allocator recycles fault inside the ROCm runtime - tools/repro_pin_fault.py, profiles/r05_pin_fault.txt.) */
void *ecl_hip_alloc_host(size_t bytes);
void ecl_hip_free_host(void *p);

/* ==== 3. TUNING: none of these changes a result ==== */
/* Optional: allocate now what a later ecl_hip_add_range of `nkeys` keys with record capacity `cap` will need (table,
   lane centres, prefix-product chains, record buffer), so that the first call does not pay for it.  The reference has
   no counterpart (its per-thread buffers live on the stack, main.c:350-352). */
int ecl_hip_reserve(ecl_hip *h, uint64_t nkeys, uint32_t cap);

/* Optional: set up now what a later ecl_hip_mul_batch of up to n scalars with record capacity `cap` needs (the window
   table of the width in force, device staging, record buffer), so that the first batch does not pay for it - the
   counterpart of ecl_hip_reserve for `mul`; the reference builds its table at the start of cmd_mul (main.c:543). */
int ecl_hip_reserve_mul(ecl_hip *h, uint32_t n, uint32_t cap);

/* Look-ahead over small contiguous jobs.  The reference's scheduler hands out jobs of 2^21 keys (MAX_JOB_SIZE, main.c:16,418-431): 0.17 ms
   of work for this GPU, where a launch needs 2^28 keys and more to reach the kernel's rate.  When the ecl_hip_add_range calls of the
   contexts that share a filter (same flags, stride and filter contents, any device) form that pattern - equal sizes, each starting where
   the one before ended - a call that finds nothing prepared runs ONE sweep of up to `max_keys` keys from its start, keeps the sweep's hit
   records on the host and the following calls are answered from them without a launch: each call still receives exactly the hits of its
   own keys, with key_offset counted from its own start.  Default 2^30 keys (environment ECL_HIP_LOOKAHEAD_LOG2=N, 0 = off);
   max_keys = 0 turns it off for this context, else 2^22 <= max_keys <= 2^32.  Calls larger than max_keys / 4, contexts with a
   caller-set geometry (ecl_hip_set_geometry) and filters changed on the device (ecl_hip_bloom_insert) are never looked ahead for. */
int ecl_hip_set_lookahead(ecl_hip *h, uint64_t max_keys);
/* Optional hint: the scalar at which the caller's scan stops handing out jobs (ctx->range_e: a worker stops once range_s >= range_e,
   main.c:420-423); NULL withdraws it.  A sweep then never passes the job that contains `end` - nothing is computed that will not be asked
   for - and starts with the second contiguous job; without the hint a sweep covers at most half of what the pattern has consumed so far.
   Either way a sweep is only run if it replaces at least 4 jobs per context of the group (N equal shards handed to N GPUs stay N launches). */
int ecl_hip_set_scan_end(ecl_hip *h, const uint64_t end[4]);
/* Measurement: sweeps this context has run and their keys; calls answered from a sweep (this context's or a sharing one's) and their keys. */
int ecl_hip_get_lookahead_stats(ecl_hip *h, uint64_t *sweeps, uint64_t *swept_keys, uint64_t *served_calls, uint64_t *served_keys);

/* Optional: fix the window width of this context's `mul` table (8..29 bits; 0 = automatic, the default) from the next
   ecl_hip_mul_batch on - a caller that knows it will multiply billions of scalars takes 22 at once.  No reference
   counterpart other than the compile-time _GTABLE_W (lib/ecc.c:876). */
int ecl_hip_set_mul_window(ecl_hip *h, uint32_t bits);
/* ... and the width of the table the context holds right now (0: none yet). */
int ecl_hip_get_mul_window(ecl_hip *h, uint32_t *bits);

/* Geometry of the walk: half_group = table points per group (the reference fixes 1024: GROUP_INV_SIZE/2,
   main.c:17), max_lanes = keys walked concurrently.  0 keeps the default (1024 and 2^21 lanes; the walk parks
   lanes * half_group * 36 bytes of prefix products in HBM - 77 GB at the default - and takes fewer lanes by itself
   when free memory is short; while half_group is left at its default, calls shorter than lanes * 2048 keys use a half
   group of 512 or 256 so that the lanes stay oversubscribed).  Results do not depend on either. */
int ecl_hip_set_geometry(ecl_hip *h, uint32_t half_group, uint32_t max_lanes);
/* Current geometry.  One "sweep" = lanes * 2 * half_group keys: calls whose nkeys is a multiple of it keep every
   lane equally busy (no tail) and can be continued by the next contiguous call without re-initialisation. */
int ecl_hip_get_geometry(ecl_hip *h, uint32_t *half_group, uint32_t *lanes);

/* The geometry ecl_hip_add_range would walk a call of `nkeys` keys with: half group, lanes, groups per lane.  While the half group is
   left automatic it follows a measured cost model (one inversion per lane and group against keeping the chip oversubscribed):
   8 x 2^17 lanes for the reference's 2^21-key job, 32 x 2^18 for 2^24 keys, 128 x 2^21 for 2^29, 1024 x 2^21 for 2^32.  Contiguous
   calls continue the resident walk only while this stays the same, i.e. for calls of one size. */
int ecl_hip_plan_geometry(ecl_hip *h, uint64_t nkeys, uint32_t *half_group, uint32_t *lanes, uint32_t *groups_per_lane);

/* ==== 4. MEASUREMENT ==== */
/* Measurement: accumulated HIP-event time of the main add kernel since the last reset, and launch count. */
int ecl_hip_get_timing(ecl_hip *h, double *kernel_ms, uint64_t *launches, uint64_t *keys);
int ecl_hip_reset_timing(ecl_hip *h);
/* ... of the set-up a non-contiguous ecl_hip_add_range pays before its search kernel (base centre through the window
   table, lane centres; HIP events on the handle's stream), and how many calls paid it; contiguous follow-up calls
   pay nothing.  Reset by ecl_hip_reset_timing. */
int ecl_hip_get_setup_timing(ecl_hip *h, double *setup_ms, uint64_t *setups);
/* ... of ecl_hip_mul_batch: HIP-event time from the first chunk's copy being awaited to the last kernel (host->device
   copies overlapped with the kernels), calls and scalars since the last reset. */
int ecl_hip_get_mul_timing(ecl_hip *h, double *ms, uint64_t *calls, uint64_t *scalars);

/* Known-answer test of the device code (hash160 of 1*G, 2*G, 0xdc2a04*G, both encodings, via the double-and-add
   kernel) and a cross-check of the walk kernel against it over 4096 consecutive keys.  ecl_hip_open() runs it
   (a few ms) unless the environment has ECL_HIP_SKIP_SELFTEST=1; a failure makes open return ECL_E_SELFTEST. */
int ecl_hip_selftest(ecl_hip *h);

/* ==== 5. DIAGNOSTICS: device primitives exposed for the parity tests (each runs a tiny kernel); not for production callers ==== */
/* op: 0 mul, 1 sqr, 2 inv (the one the kernels use), 3 sub, 4 add, 5 neg, 6 sqr(sqr a), 7 (a*b)*b, 8 sqr(a)*a, 9 inv by division steps,
   10 inv by the addition chain of lib/ecc.c:463-520, 11 1 / (4b - 2a) by division steps from an unnormalised operand; a,b,r: n field
   elements as 4 little-endian u64 limbs */
int ecl_hip_diag_fe(ecl_hip *h, int op, const uint64_t (*a)[4], const uint64_t (*b)[4], uint64_t (*r)[4], uint32_t n);
/* affine public keys of n scalars (double-and-add kernel); ok[i] = 0 for the point at infinity */
int ecl_hip_diag_mulg(ecl_hip *h, const uint64_t (*k)[4], uint64_t (*x)[4], uint64_t (*y)[4], uint8_t *ok, uint32_t n);
/* hash160 of n affine points, both encodings */
int ecl_hip_diag_hash160(ecl_hip *h, const uint64_t (*x)[4], const uint64_t (*y)[4], uint32_t (*h33)[5],
                         uint32_t (*h65)[5], uint32_t n);
/* bloom probe of n hashes against the resident filter */
int ecl_hip_diag_bloom(ecl_hip *h, const uint32_t (*h160)[5], uint8_t *hit, uint32_t n);
/* the probe's word index r[i] = x[i] mod nwords (the `% (blf->size * 64)` of utils.c:286-288 on the word index) as the
   device computes it, for any filter size 0 < nwords < 2^58 and x < 2^58; no filter needs to be resident */
int ecl_hip_diag_bloom_mod(ecl_hip *h, uint64_t nwords, const uint64_t *x, uint64_t *r, uint32_t n);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif
