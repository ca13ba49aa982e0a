// abi_lookahead_ctx.h — what a context carries for the look-ahead of abi_lookahead.h (included before `struct ecl_hip`; the CPU test
// harness csrc/tools/lookahead_host.cpp builds its stand-in context from the same list, so the logic it tests is the library's own).
#pragma once
struct la_group;
struct la_region;
// the group of contexts this one shares sweeps with; the sweep the last call was answered from (ecl_hip_fetch_found reads the rest of
// that call's records there); the caller's scan end; counters
#define LA_CONTEXT_MEMBERS                                                                                                          \
  u64 la_max = 0;            /* keys per sweep at most; 0 = off */                                                                  \
  bool geom_fixed = false;   /* the caller set a walk geometry: it wants its calls launched as given */                             \
  std::shared_ptr<struct la_group> grp;                                                                                             \
  bool la_key_valid = false; /* the filter on the device is the one la_bloom_fp describes */                                        \
  u64 la_bloom_fp = 0, la_list_fp = 0;                                                                                              \
  bool la_have_end = false;                                                                                                         \
  u256 la_end;                                                                                                                      \
  std::shared_ptr<struct la_region> last_region;                                                                                    \
  size_t last_host_at = 0;                                                                                                          \
  u32 last_host_n = 0;                                                                                                              \
  u64 last_host_off = 0;                                                                                                            \
  bool last_from_host = false;                                                                                                      \
  std::vector<ecl_found> la_buf;                                                                                                    \
  uint64_t la_sweeps = 0, la_swept_keys = 0, la_served_calls = 0, la_served_keys = 0;
