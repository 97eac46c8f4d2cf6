// First-use selection among kernel variants that produce identical results: every candidate is timed on the caller's own
// arguments (the output is simply written several times), on a warmed-up chip and in two passes, and the fastest is remembered per
// problem shape.
// Falls back to the caller's cost model when FN2_AUTOTUNE=0, or while the stream is being captured into a graph (no host
// synchronisation allowed there) for a shape that has no remembered pick yet.
#pragma once
#include <array>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>

#include "fn2_common.hpp"

namespace fn2 {

using TuneKey = std::array<int, 10>;

// FN2_AUTOTUNE_CACHE=<file>: picks are appended to / preloaded from a text file ("<cache name> k0 .. k9 best" per line), so a
// second process (e.g. a profiled run) launches no candidates.
struct TuneCache {
  // table_version: something that changes with the variant table (its size): a cache file written by another build never matches
  TuneCache(const char* name_, int table_version) : name(std::string(name_) + "#" + std::to_string(table_version)) {}
  std::mutex mu;
  std::map<TuneKey, int> best;
  std::string name;
  bool loaded = false;
  void load_locked() {
    if (loaded) return;
    loaded = true;
    const char* path = std::getenv("FN2_AUTOTUNE_CACHE");
    if (!path) return;
    if (FILE* f = std::fopen(path, "r")) {
      char nm[64];
      TuneKey k;
      int b;
      while (std::fscanf(f, "%63s %d %d %d %d %d %d %d %d %d %d %d", nm, &k[0], &k[1], &k[2], &k[3], &k[4], &k[5], &k[6], &k[7], &k[8], &k[9], &b) == 12)
        if (name == nm) best[k] = b;
      std::fclose(f);
    }
  }
  void store_locked(const TuneKey& k, int b) {
    best[k] = b;
    const char* path = std::getenv("FN2_AUTOTUNE_CACHE");
    if (!path) return;
    if (FILE* f = std::fopen(path, "a")) {
      std::fprintf(f, "%s %d %d %d %d %d %d %d %d %d %d %d\n", name.c_str(), k[0], k[1], k[2], k[3], k[4], k[5], k[6], k[7], k[8], k[9], b);
      std::fclose(f);
    }
  }
};

// FN2_AUTOTUNE=0 switches the selection off altogether (cost model only).  While the stream is being captured into a graph no
// candidate can be timed (no host synchronisation there), but a pick remembered from an earlier eager call is still used.
inline bool autotune_enabled(hipStream_t) {
  static const bool on = [] { const char* e = std::getenv("FN2_AUTOTUNE"); return !(e && e[0] == '0'); }();
  return on;
}

inline bool stream_is_capturing(hipStream_t st) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  return hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone;
}

// run(candidate) launches one candidate and returns FN2_OK, or a non-zero status if it does not apply (skipped).
// Returns the chosen candidate (cached), or -1 if none could be timed.
// usable(candidate): cheap host-side check that a candidate applies to this problem; a remembered pick (a FN2_AUTOTUNE_CACHE file
// can be stale or corrupt) is used only if it is in range and still usable -- otherwise it is dropped and the shape is tuned again.
template <class Run, class Usable>
int autotune_pick(TuneCache& cache, const TuneKey& key, int ncand, hipStream_t st, Run run, Usable usable) {
  {
    std::lock_guard<std::mutex> lk(cache.mu);
    cache.load_locked();
    auto it = cache.best.find(key);
    if (it != cache.best.end()) {
      if (it->second >= 0 && it->second < ncand && usable(it->second)) return it->second;
      cache.best.erase(it);
    }
  }
  if (stream_is_capturing(st)) return -1;
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess) return -1;
  if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); return -1; }
  auto timed = [&](int c, float& ms) -> bool {                       // two launches of candidate c between events
    (void)hipEventRecord(e0, st);
    run(c); run(c);
    (void)hipEventRecord(e1, st);
    return hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess;
  };
  // The chip needs ~25 ms of load to reach its steady clocks (2.0 -> 2.4 GHz): candidates timed one after the other from a cold start
  // are timed at different clocks, later ones up to 15 % faster.  So: (1) the first applicable candidate runs until its time has
  // settled (two consecutive readings within 2 %, at most ~40 ms), (2) every candidate is timed in TWO passes over the list and keeps its
  // better reading.
  int first = -1;
  for (int c = 0; c < ncand && first < 0; ++c) {
    if (run(c) == FN2_OK) first = c; else (void)hipGetLastError();
  }
  if (first >= 0) {
    float prev = 0.f, total = 0.f;
    for (int rep = 0; rep < 200 && total < 40.f; ++rep) {
      float ms = 0.f;
      if (!timed(first, ms)) break;
      total += ms;
      if (rep > 0 && ms > 0.98f * prev && ms < 1.02f * prev && total > 2.f) break;
      prev = ms;
    }
  }
  int best = -1;
  float best_ms = 0.f;
  for (int pass = 0; pass < 2 && first >= 0; ++pass)
    for (int c = first; c < ncand; ++c) {
      if (pass == 0 && c != first) {
        if (run(c) != FN2_OK) { (void)hipGetLastError(); continue; }  // warm-up launch (also: does it apply at all)
      } else if (pass == 1 && run(c) != FN2_OK) { (void)hipGetLastError(); continue; }
      float ms = 0.f;
      if (!timed(c, ms)) continue;
      if (best < 0 || ms < best_ms) { best = c; best_ms = ms; }
    }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  last_error().clear();
  if (best >= 0) {
    std::lock_guard<std::mutex> lk(cache.mu);
    cache.store_locked(key, best);
  }
  return best;
}

}  // namespace fn2
