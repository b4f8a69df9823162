// examples/multi_gpu.cpp -- the C++ host of the multi-GPU receive path: one process per GPU, channels sharded, the
// coefficient blob broadcast once over RCCL after a filter change, no data-path collective (BASELINE
// configs[4]).  Everything the ranks do goes through the C ABI of include/t41rx.h; nothing here knows a kernel.
//
// What it restates from the reference: CalcFilters() (Filter.cpp:235-249) rewrites the coefficient arrays in place when
// the operator changes a filter edge; with the batch spread over GPUs that happens on ONE rank (the one that owns the
// front panel) and the others must end up with the same arrays AND the same parameters:
//     rank 0 : t41rx_set_params(new edges)  ->  t41rx_get_coeffs(blob)
//     all    : ncclBroadcast(blob, root 0)                      (RCCL over xGMI; ~5 KiB, latency-bound)
//     rank r : t41rx_set_coeffs(blob)       ->  t41rx_get_params() == rank 0's
// then every rank streams its own channels (t41rx_process_device) and only the timing is reduced (ncclAllReduce max).
//
//   examples/multi_gpu [--gpus N] [--channels C] [--frames F] [--steps K] [--warmup W]
//
// The parent forks its N ranks BEFORE anything touches a GPU (a process that has initialised HIP must not fork or exec),
// hands each a pipe end for the 128-byte ncclUniqueId rank 0 creates, waits, and fails if any rank fails.  Rank r uses
// HIP device r.  Rank 0 prints one JSON line.  Each rank also checks the broadcast path against the direct one: a second
// context designed locally for the new edges must produce bit-identical audio.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/wait.h>
#include <unistd.h>

#include <csignal>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "t41rx.h"

#define HIPCHK(x)                                                                                         \
  do {                                                                                                    \
    hipError_t e_ = (x);                                                                                  \
    if (e_ != hipSuccess) {                                                                               \
      std::fprintf(stderr, "rank %d: %s -> %s\n", g_rank, #x, hipGetErrorString(e_));                     \
      return 10;                                                                                          \
    }                                                                                                     \
  } while (0)
#define NCCLCHK(x)                                                                                        \
  do {                                                                                                    \
    ncclResult_t r_ = (x);                                                                                \
    if (r_ != ncclSuccess) {                                                                              \
      std::fprintf(stderr, "rank %d: %s -> %s\n", g_rank, #x, ncclGetErrorString(r_));                    \
      return 11;                                                                                          \
    }                                                                                                     \
  } while (0)
#define RXCHK(x)                                                                                          \
  do {                                                                                                    \
    int s_ = (x);                                                                                         \
    if (s_ != T41RX_OK) {                                                                                 \
      std::fprintf(stderr, "rank %d: %s -> %s (%s)\n", g_rank, #x, t41rx_strerror(s_), t41rx_last_error()); \
      return 12;                                                                                          \
    }                                                                                                     \
  } while (0)

static int g_rank = -1;

struct Opts {
  int gpus = 1, channels = 4096, frames = 8, steps = 10, warmup = 3;
};

// contiguous channel range of a rank (sizes differ by at most one) -- t41_sdr_amd/dist.py: shard_channels
static void shard(long total, int rank, int world, long *lo, long *hi) {
  const long base = total / world, rem = total % world;
  *lo = rank * base + (rank < rem ? rank : rem);
  *hi = *lo + base + (rank < rem ? 1 : 0);
}

static bool read_all(int fd, void *buf, size_t n) {
  char *p = static_cast<char *>(buf);
  while (n > 0) {
    const ssize_t k = read(fd, p, n);
    if (k <= 0) return false;
    p += k;
    n -= (size_t)k;
  }
  return true;
}
static bool write_all(int fd, const void *buf, size_t n) {
  const char *p = static_cast<const char *>(buf);
  while (n > 0) {
    const ssize_t k = write(fd, p, n);
    if (k <= 0) return false;
    p += k;
    n -= (size_t)k;
  }
  return true;
}

// SURVEY 8d's synthetic input for one channel: three tones + noise, the first inside the USB pass band after the
// I flip, the +Fs/4 shift and the NCO (48000 - nco - audio Hz)
static void synth(std::vector<float> &I, std::vector<float> &Q, size_t off, size_t n, int nco_hz, uint64_t seed) {
  std::mt19937_64 rng(seed);
  std::uniform_real_distribution<double> u(0.0, 1.0);
  std::normal_distribution<double> g(0.0, 1.0);
  double a[3], f[3], ph[3];
  for (int k = 0; k < 3; ++k) {
    a[k] = 0.05 + 0.25 * u(rng);
    f[k] = -90000.0 + 180000.0 * u(rng);
    ph[k] = 6.283185307179586 * u(rng);
  }
  f[0] = 48000.0 - nco_hz - (400.0 + 2100.0 * u(rng));
  for (size_t i = 0; i < n; ++i) {
    double re = 0.0, im = 0.0;
    for (int k = 0; k < 3; ++k) {
      const double w = 6.283185307179586 * f[k] / 192000.0 * (double)i + ph[k];
      re += a[k] * std::cos(w);
      im += a[k] * std::sin(w);
    }
    re += 0.01 / std::sqrt(2.0) * g(rng);
    im += 0.01 / std::sqrt(2.0) * g(rng);
    I[off + i] = (float)std::fmax(-0.999, std::fmin(0.999, re));
    Q[off + i] = (float)std::fmax(-0.999, std::fmin(0.999, im));
  }
}

static int run_rank(int rank, const Opts &o, int id_fd_read, const std::vector<int> &id_fd_write) {
  g_rank = rank;
  const int world = o.gpus;
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (rank >= ndev) {
    std::fprintf(stderr, "rank %d: only %d HIP device(s) visible: one process per GPU needs %d\n", rank, ndev, world);
    return 13;
  }
  HIPCHK(hipSetDevice(rank));
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, rank));
  std::fprintf(stderr, "rank %d of %d: device %d = %s (%s)\n", rank, world, rank, prop.name, prop.gcnArchName);

  // ---- the communicator: rank 0 creates the id, the parent's pipes carry it
  ncclUniqueId id;
  if (rank == 0) {
    NCCLCHK(ncclGetUniqueId(&id));
    for (int fd : id_fd_write)
      if (!write_all(fd, &id, sizeof(id))) return 14;
  } else if (!read_all(id_fd_read, &id, sizeof(id))) {
    return 14;
  }
  ncclComm_t comm;
  NCCLCHK(ncclCommInitRank(&comm, world, id, rank));
  hipStream_t stream;
  HIPCHK(hipStreamCreate(&stream));

  // ---- contexts: every rank starts from the defaults (20 m: USB 200..3000 Hz); this rank's shard of the batch
  long lo, hi;
  shard((long)world * o.channels, rank, world, &lo, &hi);
  const int nch = (int)(hi - lo);
  t41rx_params p0;
  t41rx_default_params(&p0);
  t41rx_ctx *rx = nullptr;
  RXCHK(t41rx_create(&rx, rank, nch, &p0));
  const int L = t41rx_frame_len(rx);
  std::vector<int32_t> nco(nch);
  for (int c = 0; c < nch; ++c) nco[c] = (int32_t)(-43000 + 50 * (int)(((lo + c) * 2654435761ull) % 1661));  // [-43000, 40000] Hz
  RXCHK(t41rx_set_nco_freq(rx, nco.data(), nch));

  // ---- the filter change, on rank 0 only (CalcFilters(), Filter.cpp:235-249) ...
  t41rx_params p1 = p0;
  p1.FLoCut = 300;
  p1.FHiCut = 2700;
  p1.audioVolume = 40;
  const size_t blob_bytes = t41rx_coeff_blob_bytes(p0.fft_length);
  std::vector<unsigned char> blob(blob_bytes);
  if (rank == 0) {
    RXCHK(t41rx_set_params(rx, &p1));
    RXCHK(t41rx_get_coeffs(rx, blob.data(), blob_bytes));
  }
  // ---- ... and its one collective: the blob, root 0 -> everyone, on device buffers (RCCL moves device memory)
  unsigned char *d_blob = nullptr;
  HIPCHK(hipMalloc((void **)&d_blob, blob_bytes));
  if (rank == 0) HIPCHK(hipMemcpyAsync(d_blob, blob.data(), blob_bytes, hipMemcpyHostToDevice, stream));
  NCCLCHK(ncclBroadcast(d_blob, d_blob, blob_bytes, ncclChar, 0, comm, stream));
  HIPCHK(hipMemcpyAsync(blob.data(), d_blob, blob_bytes, hipMemcpyDeviceToHost, stream));
  HIPCHK(hipStreamSynchronize(stream));
  if (rank != 0) RXCHK(t41rx_set_coeffs(rx, blob.data(), blob_bytes));
  t41rx_params got;
  RXCHK(t41rx_get_params(rx, &got));
  if (std::memcmp(&got, &p1, sizeof(got)) != 0) {
    std::fprintf(stderr, "rank %d: parameters after the broadcast differ from rank 0's (FLoCut %d FHiCut %d)\n", rank, got.FLoCut, got.FHiCut);
    return 15;
  }

  // ---- this rank's input: [nch][frames * L] planar f32, resident in HBM before the timed region
  const size_t per = (size_t)o.frames * (size_t)L, total = (size_t)nch * per;
  std::vector<float> hI(total), hQ(total);
  for (int c = 0; c < nch; ++c) synth(hI, hQ, (size_t)c * per, per, nco[c], 0x5441315Full + (uint64_t)(lo + c));
  float *dI, *dQ, *dOut, *dRef;
  HIPCHK(hipMalloc((void **)&dI, total * 4));
  HIPCHK(hipMalloc((void **)&dQ, total * 4));
  HIPCHK(hipMalloc((void **)&dOut, total * 4));
  HIPCHK(hipMalloc((void **)&dRef, total * 4));
  HIPCHK(hipMemcpy(dI, hI.data(), total * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dQ, hQ.data(), total * 4, hipMemcpyHostToDevice));

  // ---- the broadcast path against the direct one: a context designed HERE for the new edges, same stream of samples
  {
    t41rx_ctx *direct = nullptr;
    RXCHK(t41rx_create(&direct, rank, nch, &p1));
    RXCHK(t41rx_set_nco_freq(direct, nco.data(), nch));
    RXCHK(t41rx_process_device(direct, dI, dQ, dRef, o.frames, stream));
    RXCHK(t41rx_process_device(rx, dI, dQ, dOut, o.frames, stream));
    HIPCHK(hipStreamSynchronize(stream));
    std::vector<float> a(total), b(total);
    HIPCHK(hipMemcpy(a.data(), dOut, total * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(b.data(), dRef, total * 4, hipMemcpyDeviceToHost));
    double peak = 0.0;
    for (size_t i = 0; i < total; ++i) peak = std::fmax(peak, std::fabs((double)a[i]));
    if (std::memcmp(a.data(), b.data(), total * 4) != 0 || !(peak > 1e-4) || !std::isfinite(peak)) {
      std::fprintf(stderr, "rank %d: audio through the broadcast coefficients differs from a locally designed context (peak %g)\n", rank, peak);
      return 16;
    }
    RXCHK(t41rx_destroy(direct));
    RXCHK(t41rx_reset(rx));
  }

  // ---- W untimed + K timed launches; ranks meet before and after (a 1-element all-reduce is the barrier)
  float *d_t = nullptr;
  HIPCHK(hipMalloc((void **)&d_t, sizeof(float)));
  HIPCHK(hipMemset(d_t, 0, sizeof(float)));
  for (int k = 0; k < o.warmup; ++k) RXCHK(t41rx_process_device(rx, dI, dQ, dOut, o.frames, stream));
  NCCLCHK(ncclAllReduce(d_t, d_t, 1, ncclFloat, ncclMax, comm, stream));
  HIPCHK(hipStreamSynchronize(stream));
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0));
  HIPCHK(hipEventCreate(&e1));
  HIPCHK(hipEventRecord(e0, stream));
  for (int k = 0; k < o.steps; ++k) RXCHK(t41rx_process_device(rx, dI, dQ, dOut, o.frames, stream));
  HIPCHK(hipEventRecord(e1, stream));
  HIPCHK(hipStreamSynchronize(stream));
  float ms = 0.0f;
  HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  HIPCHK(hipMemcpy(d_t, &ms, sizeof(float), hipMemcpyHostToDevice));
  NCCLCHK(ncclAllReduce(d_t, d_t, 1, ncclFloat, ncclMax, comm, stream));  // the slowest rank's time
  HIPCHK(hipStreamSynchronize(stream));
  float ms_max = 0.0f;
  HIPCHK(hipMemcpy(&ms_max, d_t, sizeof(float), hipMemcpyDeviceToHost));

  if (rank == 0) {
    const double samples = (double)world * (double)o.channels * (double)o.frames * (double)L * (double)o.steps;
    std::printf("{\"ok\": true, \"host\": \"C++ (examples/multi_gpu.cpp)\", \"n_gpus\": %d, \"channels_per_gpu\": %d, \"frames_per_launch\": %d, "
                "\"steps\": %d, \"warmup\": %d, \"ms_per_step\": %.5f, \"value\": %.1f, \"unit\": \"MSamples/s\", \"scaling\": \"weak\", "
                "\"coeff_blob_bytes\": %zu, \"collective\": \"ncclBroadcast of the coefficient blob (root 0) + ncclAllReduce(max) of the timing\", "
                "\"broadcast_equals_local_design\": true, \"FLoCut\": %d, \"FHiCut\": %d}\n",
                world, o.channels, o.frames, o.steps, o.warmup, ms_max / o.steps, samples / (ms_max * 1e-3) / 1e6, blob_bytes, got.FLoCut, got.FHiCut);
    std::fflush(stdout);
  }
  RXCHK(t41rx_destroy(rx));
  (void)hipFree(dI);
  (void)hipFree(dQ);
  (void)hipFree(dOut);
  (void)hipFree(dRef);
  (void)hipFree(d_blob);
  (void)hipFree(d_t);
  NCCLCHK(ncclCommDestroy(comm));
  return 0;
}

int main(int argc, char **argv) {
  Opts o;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    auto val = [&](int *dst) {
      if (i + 1 >= argc) { std::fprintf(stderr, "%s needs a value\n", a.c_str()); std::exit(2); }
      *dst = std::atoi(argv[++i]);
    };
    if (a == "--gpus") val(&o.gpus);
    else if (a == "--channels") val(&o.channels);
    else if (a == "--frames") val(&o.frames);
    else if (a == "--steps") val(&o.steps);
    else if (a == "--warmup") val(&o.warmup);
    else { std::fprintf(stderr, "usage: multi_gpu [--gpus N] [--channels C] [--frames F] [--steps K] [--warmup W]\n"); return 2; }
  }
  if (o.gpus < 1 || o.gpus > 64 || o.channels < 1 || o.frames < 1 || o.steps < 1 || o.warmup < 0) {
    std::fprintf(stderr, "bad arguments\n");
    return 2;
  }
  setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);  // dmabuf IPC (the host driver's only mode here); keeps a caller's own setting
  // pipes for the ncclUniqueId: rank 0 -> rank r, created before the fork; no HIP call has been made in this process
  std::vector<int> rd(o.gpus, -1), wr;
  for (int r = 1; r < o.gpus; ++r) {
    int fd[2];
    if (pipe(fd) != 0) { std::perror("pipe"); return 3; }
    rd[r] = fd[0];
    wr.push_back(fd[1]);
  }
  std::vector<pid_t> kids;
  for (int r = 0; r < o.gpus; ++r) {
    const pid_t pid = fork();
    if (pid < 0) { std::perror("fork"); return 3; }
    if (pid == 0) {
      const int rc = run_rank(r, o, rd[r], r == 0 ? wr : std::vector<int>());
      std::fflush(stdout);
      std::fflush(stderr);
      _exit(rc);
    }
    kids.push_back(pid);
  }
  int rc = 0;
  std::vector<pid_t> alive = kids;
  while (!alive.empty()) {
    int st = 0;
    const pid_t pid = waitpid(-1, &st, 0);
    if (pid < 0) { std::perror("waitpid"); return 3; }
    for (size_t i = 0; i < alive.size(); ++i)
      if (alive[i] == pid) { alive.erase(alive.begin() + (long)i); break; }
    if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) {
      if (rc == 0) rc = WIFEXITED(st) ? WEXITSTATUS(st) : 1;
      for (pid_t other : alive) kill(other, SIGTERM);  // they would wait at a collective forever (exact PIDs this process started, still running)
    }
  }
  return rc;
}
