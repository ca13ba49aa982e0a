/*
 * orc.h — CPU ORACLE for the ecloop `add`/`mul` hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the algorithm of vladkens/ecloop v0.5.0 (reference tree
 * /root/reference; every function below cites the reference file:line it follows).  It exists so
 * that tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg can check the HIP path
 * bit-for-bit.  Nothing in the shipped product path (ecloop_amd/, include/, the host CLI) may
 * include, link, import or execute anything from this directory.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this restatement against
 *   (a) golden vectors produced by the unmodified reference binary (oracle/_ref, built by
 *       oracle/Makefile from /root/reference; generator: tests/golden/make_golden.py), and
 *   (b) the reference's own known-answer flows (`make add` = 9 keys, `make mul` = 1080 keys,
 *       CI smoke range, G hash160 KATs) — SURVEY.md §4 / §8c.
 */
#ifndef ORC_H
#define ORC_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint64_t orc_fe[4]; /* 256-bit little-endian limbs (ecc.c:26) */
typedef struct orc_pt { orc_fe x, y, z; } orc_pt; /* homogeneous projective, z=1 when affine (ecc.c:546) */

/* --- field mod p (ecc.c:269-540) --- */
void orc_fp_add(orc_fe r, const orc_fe a, const orc_fe b);
void orc_fp_sub(orc_fe r, const orc_fe a, const orc_fe b);
void orc_fp_neg(orc_fe r, const orc_fe a);
void orc_fp_mul(orc_fe r, const orc_fe a, const orc_fe b);
void orc_fp_sqr(orc_fe r, const orc_fe a);
void orc_fp_inv(orc_fe r, const orc_fe a);
void orc_fp_grpinv(orc_fe *r, uint32_t n);

/* --- scalars mod n (ecc.c:166-265) --- */
void orc_sn_add(orc_fe r, const orc_fe a, const orc_fe b);
void orc_sn_sub(orc_fe r, const orc_fe a, const orc_fe b);
void orc_sn_neg(orc_fe r, const orc_fe a);
void orc_sn_mul(orc_fe r, const orc_fe a, const orc_fe b);
void orc_sn_add_stride(orc_fe r, const orc_fe base, const orc_fe stride, uint64_t off);
void orc_sn_from_hex(orc_fe r, const char *hex);

/* --- curve (ecc.c:611-929) --- */
void orc_pt_dbl(orc_pt *r, const orc_pt *p);
void orc_pt_add(orc_pt *r, const orc_pt *p, const orc_pt *q);
void orc_pt_rdc(orc_pt *r, const orc_pt *a);
void orc_pt_mul(orc_pt *r, const orc_pt *p, const orc_fe k);
void orc_pt_mulg_affine(orc_fe x, orc_fe y, const orc_fe k); /* ec_jacobi_mulrdc(&G1, k) */
int orc_pt_on_curve(const orc_pt *p);
size_t orc_gtable_init(void);
void orc_gtable_mul(orc_pt *r, const orc_fe k);
void orc_pt_grprdc(orc_pt *r, uint64_t n);

/* --- hash160 (addr.c, sha256.c, rmd160.c, rmd160s.c) --- */
void orc_sha256_blocks(uint32_t state[8], const uint8_t *data, uint32_t len);
void orc_rmd160_block(uint32_t out[5], const uint32_t x[16]);
void orc_hash160_33(uint32_t h[5], const orc_fe x, const orc_fe y);
void orc_hash160_65(uint32_t h[5], const orc_fe x, const orc_fe y);

/* --- bloom filter (utils.c:274-326) --- */
void orc_blf_add(uint64_t *bits, uint64_t size_words, const uint32_t h[5]);
int orc_blf_has(const uint64_t *bits, uint64_t size_words, const uint32_t h[5]);
void orc_blf_add_many(uint64_t *bits, uint64_t size_words, const uint32_t *h, uint64_t n);
void orc_blf_has_many(const uint64_t *bits, uint64_t size_words, const uint32_t *h, uint64_t n, uint8_t *hit);
uint64_t orc_blf_gen_many(uint64_t *bits, uint64_t size_words, const uint32_t *h, uint64_t n); /* utils.c:455-470 */
uint64_t orc_blf_gen_size(uint64_t n); /* utils.c:421-427 size formula (words) */

/* --- filter = bloom + optional sorted list (main.c:71-131, 205-217) --- */
typedef struct orc_filter {
  uint64_t *bits;      /* bloom words */
  uint64_t size;       /* bloom size in 64-bit words */
  uint32_t *list;      /* sorted unique 5-word hashes, or NULL for bloom-only mode */
  uint64_t list_count;
} orc_filter;
/* build list-mode filter from n unsorted 5-word hashes (copies) */
void orc_filter_from_list(orc_filter *f, const uint32_t *hashes, uint64_t n);
void orc_filter_from_bloom(orc_filter *f, const uint64_t *bits, uint64_t size_words);
void orc_filter_free(orc_filter *f);
int orc_filter_check(const orc_filter *f, const uint32_t h[5]);

/* --- found record, in the order the reference's single-thread run emits them --- */
typedef struct orc_found {
  uint32_t h160[5];
  uint8_t compressed; /* 1 = addr33, 0 = addr65 */
  uint8_t endo;       /* 0..5 (main.c:267-276) */
  uint8_t pad[2];
  uint64_t pk[4];     /* private key, little-endian limbs */
} orc_found;

typedef struct orc_add_cfg {
  int check33, check65, use_endo;
  uint32_t ord_offs;     /* stride = 2^ord_offs (main.c:221-222) */
  int verify;            /* re-derive every hit like pk_verify_hash (main.c:248-263); mismatch -> return -2 */
  int threads;           /* >=1; with >1 the found order is nondeterministic, like the reference */
  int rnd_jobs;          /* 1: one window of cmd_rnd - job_size is MAX_JOB_SIZE whatever the range (main.c:624) */
} orc_add_cfg;

/* cmd_add (main.c:405-454) over [range_s, range_e): returns 0, -1 on found overflow, -2 on verify mismatch.
   *checked receives the status-line counter (job_size per job, x6 with endo: main.c:431);
   *hashed receives the number of base keys actually hashed (ceil(job/2048)*2048 per job). */
int orc_add_range(const orc_add_cfg *cfg, const orc_filter *flt, const orc_fe range_s, const orc_fe range_e,
                  orc_found *out, uint64_t cap, uint64_t *nout, uint64_t *checked, uint64_t *hashed);

/* cmd_mul worker body (main.c:486-540) for n already-parsed scalars, batch size 1 semantics. */
int orc_mul_batch(int check33, int check65, const orc_filter *flt, const orc_fe *pk, uint64_t n, orc_found *out,
                  uint64_t cap, uint64_t *nout);

/* cmd_mul at scale: hash160 of k[i]*G for every scalar, in input order, the way the reference's workers get them (2048-scalar jobs:
   ec_gtable_mul each, one ec_jacobi_grprdc per job, addr33/addr65; main.c:486-540).  h33 / h65: n x 5 words or NULL; ok[i] = 0 for a
   scalar that is 0 (mod n) (no point; hashes zeroed).  `threads` workers pull jobs from one counter. */
void orc_mul_hash160_many(const orc_fe *pk, uint64_t n, uint32_t *h33, uint32_t *h65, uint8_t *ok, int threads);

/* calc_priv (main.c:267-276) */
void orc_calc_priv(orc_fe pk, const orc_fe start, const orc_fe stride, uint64_t off, uint8_t endo);

#ifdef __cplusplus
}
#endif
#endif
