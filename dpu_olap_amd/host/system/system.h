// system.h -- run-time configuration from the environment, the counterpart of the
// reference's host/system/system.h:7-20: SF (scale factor), MAX_THREADS, and NR_GPUS in
// place of NR_DPUS.
#pragma once

#include <cstdlib>
#include <string>
#include <thread>

namespace variables {

inline int env_int(const char* name, int fallback) {
  const char* v = std::getenv(name);
  return v && *v ? std::atoi(v) : fallback;
}

inline int max_gpus() { return env_int("NR_GPUS", 1); }

// The reference defaults SF to the device count (2048 DPUs); a GPU build defaults to a
// size that fits host memory for the Arrow-resident benchmarks.
inline int scale_factor() { return env_int("SF", 4); }

inline int max_threads() { return env_int("MAX_THREADS", (int)std::thread::hardware_concurrency()); }

inline int max_threads_no_ht() {
  return env_int("MAX_THREADS", std::max(1, (int)std::thread::hardware_concurrency() / 2));
}

}  // namespace variables
