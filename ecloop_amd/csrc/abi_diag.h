// abi_diag.h - host side of the diagnostics entry points, the bulk bloom insert and the self-test.
// (one translation unit: included by ecloop_hip.hip last)
#pragma once
// ------------------------------------------------------------------------------------------------ diagnostics (host)

extern "C" int ecl_hip_diag_fe(ecl_hip* h, int op, const uint64_t (*a)[4], const uint64_t (*b)[4], uint64_t (*r)[4],
                               uint32_t n) {
  if (!h || !a || !r || n == 0 || op < 0 || op > 11) return ECL_E_ARG;
  HIPCHK(h, hipSetDevice(h->dev));
  size_t bytes = (size_t)n * 32;
  dbuf<u32> da, db, dr;
  HIPCHK(h, hipMalloc(&da.p, bytes));
  HIPCHK(h, hipMalloc(&db.p, bytes));
  HIPCHK(h, hipMalloc(&dr.p, bytes));
  HIPCHK(h, hipMemcpy(da.p, a, bytes, hipMemcpyHostToDevice));  // little-endian u64 limbs == u32 word pairs
  HIPCHK(h, hipMemcpy(db.p, b ? b : a, bytes, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_diag_fe, dim3((n + 63) / 64), dim3(64), 0, h->stream, op, da.p, db.p, dr.p, n);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipMemcpy(r, dr.p, bytes, hipMemcpyDeviceToHost));
  return ECL_OK;
}

extern "C" int ecl_hip_diag_mulg(ecl_hip* h, const uint64_t (*k)[4], uint64_t (*x)[4], uint64_t (*y)[4], uint8_t* ok,
                                 uint32_t n) {
  if (!h || !k || !x || !y || n == 0) return ECL_E_ARG;
  HIPCHK(h, hipSetDevice(h->dev));
  dbuf<u32> dk, dout;
  dbuf<u8> dok;
  HIPCHK(h, hipMalloc(&dk.p, (size_t)n * 32));
  HIPCHK(h, hipMalloc(&dout.p, (size_t)n * 64));
  HIPCHK(h, hipMalloc(&dok.p, n));
  HIPCHK(h, hipMemcpy(dk.p, k, (size_t)n * 32, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_mul_g, dim3((n + 63) / 64), dim3(64), 0, h->stream, dk.p, dout.p, dok.p, n);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipStreamSynchronize(h->stream));
  std::vector<u32> o((size_t)n * 16);
  std::vector<u8> okv(n);
  HIPCHK(h, hipMemcpy(o.data(), dout.p, (size_t)n * 64, hipMemcpyDeviceToHost));
  HIPCHK(h, hipMemcpy(okv.data(), dok.p, n, hipMemcpyDeviceToHost));
  for (u32 i = 0; i < n; ++i) {
    memcpy(x[i], &o[(size_t)i * 16], 32);
    memcpy(y[i], &o[(size_t)i * 16 + 8], 32);
    if (ok) ok[i] = okv[i];
  }
  return ECL_OK;
}

extern "C" int ecl_hip_diag_hash160(ecl_hip* h, const uint64_t (*x)[4], const uint64_t (*y)[4], uint32_t (*h33)[5],
                                    uint32_t (*h65)[5], uint32_t n) {
  if (!h || !x || !y || !h33 || !h65 || n == 0) return ECL_E_ARG;
  HIPCHK(h, hipSetDevice(h->dev));
  dbuf<u32> dx, dy, d33, d65;
  HIPCHK(h, hipMalloc(&dx.p, (size_t)n * 32));
  HIPCHK(h, hipMalloc(&dy.p, (size_t)n * 32));
  HIPCHK(h, hipMalloc(&d33.p, (size_t)n * 20));
  HIPCHK(h, hipMalloc(&d65.p, (size_t)n * 20));
  HIPCHK(h, hipMemcpy(dx.p, x, (size_t)n * 32, hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy(dy.p, y, (size_t)n * 32, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_diag_hash, dim3((n + 63) / 64), dim3(64), 0, h->stream, dx.p, dy.p, d33.p, d65.p, n);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipMemcpy(h33, d33.p, (size_t)n * 20, hipMemcpyDeviceToHost));
  HIPCHK(h, hipMemcpy(h65, d65.p, (size_t)n * 20, hipMemcpyDeviceToHost));
  return ECL_OK;
}

extern "C" int ecl_hip_bloom_insert(ecl_hip* h, const uint32_t (*h160)[5], uint64_t n) {
  if (!h || (!h160 && n)) return ECL_E_ARG;
  if (!h->d_bloom) return ECL_E_NOBLOOM;
  if (n == 0) return ECL_OK;
  la_leave(h), h->la_key_valid = false;  // the resident filter is no longer the one that was uploaded: no shared look-ahead on it
  HIPCHK(h, hipSetDevice(h->dev));
  const u64 chunk = 1ull << 24;  // 320 MB of hashes per upload
  dbuf<u32> dh;
  HIPCHK(h, hipMalloc(&dh.p, (size_t)(n < chunk ? n : chunk) * 20));
  for (u64 at = 0; at < n; at += chunk) {
    u64 m = n - at < chunk ? n - at : chunk;
    HIPCHK(h, hipMemcpy(dh.p, h160 + at, (size_t)m * 20, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_bloom_insert, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, h->stream,
                       bloom_make(h->d_bloom, h->bloom_words), h->d_bloom, dh.p, m);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));
  }
  return ECL_OK;
}

extern "C" int ecl_hip_bloom_insert_count(ecl_hip* h, const uint32_t (*h160)[5], uint64_t n, uint64_t* added) {
  if (!h || (!h160 && n) || !added) return ECL_E_ARG;
  *added = 0;
  if (!h->d_bloom) return ECL_E_NOBLOOM;
  if (n == 0) return ECL_OK;
  if (h->bloom_words >= (1ull << (64 - BLF_CHUNK_LOG2 - 6))) return ECL_E_ARG;  // bit position must fit 44 bits (2 TB filter)
  la_leave(h), h->la_key_valid = false;
  HIPCHK(h, hipSetDevice(h->dev));
  const u64 chunk = 1ull << BLF_CHUNK_LOG2;
  dbuf<u32> dh;
  dbuf<u64> tab;
  dbuf<unsigned long long> cnt;
  HIPCHK(h, hipMalloc(&dh.p, (size_t)(n < chunk ? n : chunk) * 20));
  HIPCHK(h, hipMalloc(&tab.p, sizeof(u64) << BLF_TAB_LOG2));
  HIPCHK(h, hipMalloc(&cnt.p, sizeof(unsigned long long)));
  HIPCHK(h, hipMemsetAsync(cnt.p, 0, sizeof(unsigned long long), h->stream));
  const bloom_t b = bloom_make(h->d_bloom, h->bloom_words);
  for (u64 at = 0; at < n; at += chunk) {
    const u32 m = (u32)(n - at < chunk ? n - at : chunk);
    HIPCHK(h, hipMemcpyAsync(dh.p, h160 + at, (size_t)m * 20, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemsetAsync(tab.p, 0xFF, sizeof(u64) << BLF_TAB_LOG2, h->stream));
    hipLaunchKernelGGL(k_blf_claim, dim3((m + 255) / 256), dim3(256), 0, h->stream, b, dh.p, m, tab.p);
    HIPCHK(h, hipGetLastError());
    hipLaunchKernelGGL(k_blf_count_and_set, dim3((m + 255) / 256), dim3(256), 0, h->stream, b, h->d_bloom, dh.p, m, tab.p, cnt.p);
    HIPCHK(h, hipGetLastError());
    hipLaunchKernelGGL(k_bloom_insert, dim3((m + 255) / 256), dim3(256), 0, h->stream, b, h->d_bloom, dh.p, (u64)m);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));  // the host buffer slice is free again; next chunk sees these bits
  }
  unsigned long long c = 0;
  HIPCHK(h, hipMemcpy(&c, cnt.p, sizeof c, hipMemcpyDeviceToHost));
  *added = c;
  return ECL_OK;
}

extern "C" int ecl_hip_get_bloom(ecl_hip* h, uint64_t* bits, uint64_t nwords) {
  if (!h || !bits) return ECL_E_ARG;
  if (!h->d_bloom) return ECL_E_NOBLOOM;
  if (nwords != h->bloom_words) return ECL_E_ARG;
  HIPCHK(h, hipSetDevice(h->dev));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipMemcpy(bits, h->d_bloom, nwords * sizeof(u64), hipMemcpyDeviceToHost));
  return ECL_OK;
}

extern "C" int ecl_hip_diag_bloom(ecl_hip* h, const uint32_t (*h160)[5], uint8_t* hit, uint32_t n) {
  if (!h || !h160 || !hit || n == 0) return ECL_E_ARG;
  if (!h->d_bloom) return ECL_E_NOBLOOM;
  HIPCHK(h, hipSetDevice(h->dev));
  dbuf<u32> dh;
  dbuf<u8> dhit;
  HIPCHK(h, hipMalloc(&dh.p, (size_t)n * 20));
  HIPCHK(h, hipMalloc(&dhit.p, n));
  HIPCHK(h, hipMemcpy(dh.p, h160, (size_t)n * 20, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_diag_bloom, dim3((n + 63) / 64), dim3(64), 0, h->stream, bloom_make(h->d_bloom, h->bloom_words),
                     dh.p, dhit.p, n);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipMemcpy(hit, dhit.p, n, hipMemcpyDeviceToHost));
  return ECL_OK;
}

extern "C" int ecl_hip_diag_bloom_mod(ecl_hip* h, uint64_t nwords, const uint64_t* x, uint64_t* r, uint32_t n) {
  if (!h || !x || !r || n == 0 || nwords == 0 || nwords >= (1ull << 58)) return ECL_E_ARG;
  HIPCHK(h, hipSetDevice(h->dev));
  dbuf<u64> dx, dr;
  HIPCHK(h, hipMalloc(&dx.p, (size_t)n * 8));
  HIPCHK(h, hipMalloc(&dr.p, (size_t)n * 8));
  HIPCHK(h, hipMemcpy(dx.p, x, (size_t)n * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_diag_bloom_mod, dim3((n + 63) / 64), dim3(64), 0, h->stream, bloom_make(nullptr, nwords), dx.p, dr.p, n);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipMemcpy(r, dr.p, (size_t)n * 8, hipMemcpyDeviceToHost));
  return ECL_OK;
}

// ------------------------------------------------------------------------------------------------ self-test

extern "C" int ecl_hip_selftest(ecl_hip* h) {
  if (!h) return ECL_E_ARG;
  // (1) known answers: hash160 of k*G for k = 1, 2, 0xdc2a04 (compressed, uncompressed), public vectors
  static const uint64_t KS[3][4] = {{1, 0, 0, 0}, {2, 0, 0, 0}, {0xdc2a04, 0, 0, 0}};
  static const uint32_t KAT33[3][5] = {{0x751e76e8u, 0x199196d4u, 0x54941c45u, 0xd1b3a323u, 0xf1433bd6u},
                                       {112186475u, 3455918831u, 2494304810u, 2703172626u, 1151565516u},
                                       {156887041u, 569600746u, 330545875u, 1640062380u, 639147567u}};
  static const uint32_t KAT65[3][5] = {{0x91b24bf9u, 0xf5288532u, 0x960ac687u, 0xabb03512u, 0x7b1d28a5u},
                                       {3603490856u, 3253510587u, 2691031480u, 1042137763u, 1849195074u},
                                       {3514751675u, 162192179u, 1444810732u, 2475417333u, 3394525481u}};
  uint64_t x[3][4], y[3][4];
  uint8_t ok[3];
  uint32_t h33[3][5], h65[3][5];
  int rc = ecl_hip_diag_mulg(h, KS, x, y, ok, 3);
  if (rc == ECL_OK) rc = ecl_hip_diag_hash160(h, x, y, h33, h65, 3);
  if (rc != ECL_OK) return rc;
  if (memcmp(h33, KAT33, sizeof KAT33) != 0 || memcmp(h65, KAT65, sizeof KAT65) != 0 || !(ok[0] && ok[1] && ok[2])) {
    h->err = "known-answer test of k*G -> hash160 failed";
    return ECL_E_SELFTEST;
  }
  // (2) the walk kernel against the double-and-add kernel: 4096 consecutive keys through an all-ones filter
  const u32 N = 4096, saveB = h->B, saveT = h->Tmax;
  const bool saveAuto = h->B_auto;
  u64* save_bloom = h->d_bloom;
  const u64 save_words = h->bloom_words, save_list_n = h->list_n;
  u32* save_list = h->d_list;
  h->d_list = nullptr, h->list_n = 0;
  std::vector<u64> ones(64, ~0ull);
  h->d_bloom = nullptr, h->bloom_words = 0;
  h->B = 16, h->Tmax = 256, h->B_auto = false;
  const uint64_t start[4] = {0x0123456789abcdefull, 0x1f, 0, 0};
  const u32 per_key = ((h->flags & ECL_ADDR33) ? 1 : 0) + ((h->flags & ECL_ADDR65) ? 1 : 0);
  const u32 cap = N * per_key * ((h->flags & ECL_ENDO) ? 6 : 1);
  std::vector<ecl_found> recs(cap);
  u32 n = 0;
  rc = ecl_hip_set_bloom(h, ones.data(), ones.size());
  if (rc == ECL_OK) rc = ecl_hip_add_range(h, start, N, recs.data(), cap, &n);
  std::vector<uint64_t> ks((size_t)N * 4), xs((size_t)N * 4), ys((size_t)N * 4);
  std::vector<uint32_t> r33((size_t)N * 5), r65((size_t)N * 5);
  const u256 s = sc_pow2(h->offs);
  u256 cur = sc_reduce(u256_from(start));
  for (u32 i = 0; i < N; ++i) {
    memcpy(&ks[(size_t)i * 4], cur.w, 32);
    cur = sc_add(cur, s);
  }
  if (rc == ECL_OK) rc = ecl_hip_diag_mulg(h, (const uint64_t(*)[4])ks.data(), (uint64_t(*)[4])xs.data(), (uint64_t(*)[4])ys.data(), nullptr, N);
  if (rc == ECL_OK) rc = ecl_hip_diag_hash160(h, (const uint64_t(*)[4])xs.data(), (const uint64_t(*)[4])ys.data(),
                                              (uint32_t(*)[5])r33.data(), (uint32_t(*)[5])r65.data(), N);
  // restore the caller's state whatever happened
  if (h->d_bloom) (void)hipFree(h->d_bloom);
  h->d_bloom = save_bloom, h->bloom_words = save_words;
  h->d_list = save_list, h->list_n = save_list_n;
  h->B = saveB, h->Tmax = saveT, h->B_auto = saveAuto;
  if (h->d_tab) (void)hipFree(h->d_tab);
  h->d_tab = nullptr, h->tab_B = 0, h->walk_valid = false;
  h->kernel_ms = 0, h->launches = 0, h->keys = 0, h->setup_ms = 0, h->setups = 0;
  if (rc != ECL_OK) return rc;
  u32 seen = 0;
  bool good = n == cap;
  for (u32 i = 0; i < n && good; ++i) {
    const ecl_found& f = recs[i];
    if (f.key_offset >= N) { good = false; break; }
    if (f.endo != 0) continue;  // the endomorphism images are covered by the parity tests; here: the walk itself
    const uint32_t* want = f.compressed ? &r33[f.key_offset * 5] : &r65[f.key_offset * 5];
    good = memcmp(f.h160, want, 20) == 0;
    ++seen;
  }
  if (!good || seen != N * per_key) {
    h->err = "walk kernel disagrees with the double-and-add kernel";
    return ECL_E_SELFTEST;
  }
  // (3) the window-table sum (gtable_mul: ecl_hip_verify, the base centre of every non-contiguous walk, `mul`) against the
  // double-and-add kernel on full-width scalars, so that every one of the 19 windows carries a digit: the walk's base
  // centre and the verification of its hits share this function and the table, and a hit shares its high digits with
  // the base centre - (2) exercises only the low windows.  Scalars: a fixed xorshift stream, plus every digit at its
  // maximum (0x3fff in all windows) and a single top-window digit.
  {
    const u32 M = 48;
    std::vector<uint64_t> vk((size_t)M * 4), vx((size_t)M * 4), vy((size_t)M * 4);
    std::vector<uint32_t> w33((size_t)M * 5), w65((size_t)M * 5), g33((size_t)M * 5), g65((size_t)M * 5);
    std::vector<uint8_t> vok(M), gok(M);
    u64 z = 0x9E3779B97F4A7C15ull;
    for (size_t i = 0; i < vk.size(); ++i) {
      z ^= z << 13, z ^= z >> 7, z ^= z << 17;
      vk[i] = z;
    }
    for (int w = 0; w < 4; ++w) vk[w] = ~0ull;                 // all digits 0x3fff (the sum is (2^256 - 1) mod n times G)
    vk[4] = 0, vk[5] = 0, vk[6] = 0, vk[7] = 1ull << 60;       // window 18 only
    rc = ecl_hip_verify(h, (const uint64_t(*)[4])vk.data(), M, (uint32_t(*)[5])g33.data(), (uint32_t(*)[5])g65.data(), gok.data());
    if (rc == ECL_OK) rc = ecl_hip_diag_mulg(h, (const uint64_t(*)[4])vk.data(), (uint64_t(*)[4])vx.data(), (uint64_t(*)[4])vy.data(), vok.data(), M);
    if (rc == ECL_OK) rc = ecl_hip_diag_hash160(h, (const uint64_t(*)[4])vx.data(), (const uint64_t(*)[4])vy.data(),
                                                (uint32_t(*)[5])w33.data(), (uint32_t(*)[5])w65.data(), M);
    if (rc != ECL_OK) return rc;
    if (g33 != w33 || g65 != w65 || gok != vok) {
      h->err = "window-table scalar multiplication disagrees with the double-and-add kernel";
      return ECL_E_SELFTEST;
    }
  }
  return ECL_OK;
}
