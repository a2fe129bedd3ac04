// Input feeder: what the reference does per example in dataset_loader/wham.py:171-226 under a
// torch DataLoader with num_workers processes -- scipy.io.wavfile.read of the mixture and of every source, one random crop
// shared by the example's files, float32 conversion, normalisation, zero pad, stack, pickle back to the trainer -- as
//   * a native reader: a pool of host threads parses the WAV files straight into caller-owned (pinned) batch buffers
//     [batch][streams][time] float32 + the valid length of every example (srf_feeder_*), no Python in the loop and no
//     inter-process copies; a batch's batch * streams files are read in parallel;
//   * one device kernel for the normalisation recipe (srf_feeder_normalize): per-row mean / unbiased std, zero pad, then the
//     mixture-std rescale of wham.py:212-217 -- the raw batch crosses PCIe once and is normalised where the forward runs.
// At the 18 k separated-seconds/s of one MI355X (cfg 2) the trainer consumes ~4600 4-second examples per second and GPU;
// the reference's DataLoader delivers a few hundred.
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "srf_common.h"

// ---------------------------------------------------------------------------------------------
// WAV parsing (RIFF / WAVE: PCM 8 / 16 / 24 / 32 bit and IEEE float 32 / 64, WAVE_FORMAT_EXTENSIBLE; first channel of
// multi-channel files is NOT taken -- like scipy the sample count is per channel and WHAM files are mono: channels != 1 is
// an error).  Values as scipy.io.wavfile.read + torch.tensor(dtype=float32) deliver them: integer PCM keeps its integer
// magnitude (no 1/32768 scaling), 8-bit PCM is unsigned.
// ---------------------------------------------------------------------------------------------
struct WavInfo {
  int rate = 0, channels = 0, bits = 0, fmt = 0;   // fmt: 1 = integer PCM, 3 = IEEE float
  long frames = 0;                                 // samples per channel
  long data_off = 0;                               // byte offset of the sample data
};

static bool wav_read_header(FILE* f, WavInfo& w, std::string& err) {
  unsigned char h[12];
  if (fread(h, 1, 12, f) != 12 || memcmp(h, "RIFF", 4) != 0 || memcmp(h + 8, "WAVE", 4) != 0) {
    err = "not a RIFF/WAVE file";
    return false;
  }
  bool have_fmt = false;
  for (;;) {
    unsigned char c[8];
    if (fread(c, 1, 8, f) != 8) break;
    const uint32_t sz = c[4] | (c[5] << 8) | (c[6] << 16) | ((uint32_t)c[7] << 24);
    if (memcmp(c, "fmt ", 4) == 0) {
      unsigned char b[40] = {0};
      const size_t n = sz < 40 ? sz : 40;
      if (fread(b, 1, n, f) != n) break;
      if (sz > n) fseek(f, (long)(sz - n), SEEK_CUR);
      int tag = b[0] | (b[1] << 8);
      w.channels = b[2] | (b[3] << 8);
      w.rate = b[4] | (b[5] << 8) | (b[6] << 16) | (b[7] << 24);
      w.bits = b[14] | (b[15] << 8);
      if (tag == 0xFFFE && sz >= 26) tag = b[24] | (b[25] << 8);   // WAVE_FORMAT_EXTENSIBLE: sub-format GUID's first word
      w.fmt = tag;
      have_fmt = true;
    } else if (memcmp(c, "data", 4) == 0) {
      if (!have_fmt) {
        err = "data chunk before fmt chunk";
        return false;
      }
      if ((w.fmt != 1 && w.fmt != 3) || w.channels < 1 || w.bits % 8 != 0 || w.bits < 8 || w.bits > 64) {
        err = "unsupported WAV encoding (tag " + std::to_string(w.fmt) + ", " + std::to_string(w.bits) + " bit)";
        return false;
      }
      w.data_off = ftell(f);
      w.frames = (long)(sz / (uint32_t)(w.channels * (w.bits / 8)));
      return true;
    } else {
      fseek(f, (long)(sz + (sz & 1)), SEEK_CUR);
    }
    if (sz & 1 && memcmp(c, "fmt ", 4) == 0) fseek(f, 1, SEEK_CUR);
  }
  err = "no data chunk";
  return false;
}

// samples [start, start + n) of a mono file -> dst (float32); returns false with `err` set on failure
static bool wav_read_range(const char* path, long start, long n, float* dst, long* frames_out, std::string& err) {
  FILE* f = fopen(path, "rb");
  if (!f) {
    err = std::string("cannot open ") + path;
    return false;
  }
  WavInfo w;
  bool ok = wav_read_header(f, w, err);
  if (ok && w.channels != 1) {
    err = "expected a mono file";
    ok = false;
  }
  if (ok) {
    if (frames_out) *frames_out = w.frames;
    if (start < 0) start = 0;
    if (start > w.frames) start = w.frames;
    if (n > w.frames - start) n = w.frames - start;
    const int bps = w.bits / 8;
    std::vector<unsigned char> buf((size_t)(n > 0 ? n : 0) * bps);
    if (n > 0) {
      fseek(f, w.data_off + start * bps, SEEK_SET);
      if (fread(buf.data(), 1, buf.size(), f) != buf.size()) {
        err = "short read";
        ok = false;
      }
    }
    if (ok) {
      const unsigned char* p = buf.data();
      for (long i = 0; i < n; ++i, p += bps) {
        float v;
        if (w.fmt == 3 && bps == 4) {
          memcpy(&v, p, 4);
        } else if (w.fmt == 3 && bps == 8) {
          double d;
          memcpy(&d, p, 8);
          v = (float)d;
        } else if (bps == 1) {
          v = (float)p[0];   // unsigned 8-bit PCM, as scipy returns it
        } else if (bps == 2) {
          v = (float)(int16_t)(p[0] | (p[1] << 8));
        } else if (bps == 3) {
          // scipy >= 1.6 returns 24-bit PCM left-justified in int32
          v = (float)(int32_t)(((uint32_t)p[0] << 8) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 24));
        } else if (bps == 4 && w.fmt == 1) {
          v = (float)(int32_t)(p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24));
        } else {
          err = "unsupported sample width";
          ok = false;
          break;
        }
        dst[i] = v;
      }
    }
  }
  fclose(f);
  if (!ok) err = std::string(path) + ": " + err;
  return ok;
}

extern "C" int srf_wav_info(const char* path, int* rate, int* channels, int* bits, long* frames) {
  SRF_CHECK_ARG(path, "srf_wav_info: null path");
  FILE* f = fopen(path, "rb");
  SRF_CHECK_ARG(f != nullptr, "srf_wav_info: cannot open %s", path);
  WavInfo w;
  std::string err;
  const bool ok = wav_read_header(f, w, err);
  fclose(f);
  SRF_CHECK_ARG(ok, "srf_wav_info: %s: %s", path, err.c_str());
  if (rate) *rate = w.rate;
  if (channels) *channels = w.channels;
  if (bits) *bits = w.bits;
  if (frames) *frames = w.frames;
  return SRF_OK;
}

extern "C" int srf_wav_read(const char* path, long start, long n, float* dst, long* frames) {
  SRF_CHECK_ARG(path && (dst || n == 0) && n >= 0, "srf_wav_read: bad arguments");
  std::string err;
  const bool ok = wav_read_range(path, start, n, dst, frames, err);
  SRF_CHECK_ARG(ok, "srf_wav_read: %s", err.c_str());
  return SRF_OK;
}

// ---------------------------------------------------------------------------------------------
// feeder
// ---------------------------------------------------------------------------------------------
static inline uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ULL;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
  return x ^ (x >> 31);
}

// One file of one example: the crop [start, start + T) (wham.py:178-186,196), zero padded when the file is shorter (:157-166),
// its valid length, and -- for the mixture (stream 0) of a normalising Dataset -- {mean, unbiased std} over what the
// reference normalises: the crop when it crops (augment and a longer file), else the WHOLE file: it truncates to T only
// after normalising (wham.py:183-191).  Shared by the worker threads and by srf_feeder_read_example.
static bool read_stream(const char* path, int stream, long start, int T, int augment, int normalize, float* dst, int* len,
                        float* stat, std::string& err) {
  long fr = 0;
  bool ok = wav_read_range(path, start, T, dst, &fr, err);
  if (!ok) {
    memset(dst, 0, sizeof(float) * (size_t)T);
    *len = 0;
    return false;
  }
  long got = fr - start;
  if (got > T) got = T;
  if (got < 0) got = 0;
  if (got < T) memset(dst + got, 0, sizeof(float) * (size_t)(T - got));
  *len = (int)got;
  if (stream != 0 || !normalize) return true;     // (no statistics pass, and no second read of a long file, when nothing uses them)
  const bool whole = !(augment && fr > T) && fr > T;
  std::vector<float> rest;
  const float* p = dst;
  long n = got;
  if (whole) {
    rest.resize((size_t)fr);
    ok = wav_read_range(path, 0, fr, rest.data(), nullptr, err);
    p = rest.data();
    n = fr;
  }
  double a = 0.0, q = 0.0;
  for (long i = 0; i < n; ++i) a += p[i];
  const double mean = n > 0 ? a / n : 0.0;
  for (long i = 0; i < n; ++i) q += (p[i] - mean) * (p[i] - mean);
  stat[0] = (float)mean;
  stat[1] = n > 1 ? (float)sqrt(q / (n - 1)) : NAN;
  return ok;
}

struct srf_feeder {
  std::vector<std::string> paths;   // [item][stream]
  std::vector<long> frames;         // [item] length of the mixture file (stream 0)
  int n_items = 0, n_streams = 0, T = 0, batch = 0, augment = 0, shuffle = 0, drop_last = 1, normalize = 1;
  int rank = 0, world = 1;          // this feeder hands out shard `rank` of every GLOBAL batch of batch * world items
  uint64_t seed = 0;
  int epoch = 0;
  std::vector<int> order;           // item order of the current epoch (identical on every rank: seed and epoch only)
  long cursor = 0;                  // first item of `order` of the next GLOBAL batch

  struct Slot {
    float* wave;
    int* len;      // [batch][n_streams]: valid samples of every stream (a source file may be shorter than its mixture)
    float* stat;   // [batch][2]: {mean, unbiased std} of the mixture over the range the reference normalises it on
    int n_valid;
    std::atomic<int> pending{0};
    std::string err;
  };
  struct Job {
    Slot* slot;
    int b, item, stream;
    long start;
  };
  std::deque<Slot*> inflight;   // submission order
  std::deque<Job> jobs;
  std::mutex mu;
  std::condition_variable cv_job, cv_done;
  std::vector<std::thread> workers;
  bool stop = false;

  void work() {
    for (;;) {
      Job j;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv_job.wait(lk, [&] { return stop || !jobs.empty(); });
        if (stop && jobs.empty()) return;
        j = jobs.front();
        jobs.pop_front();
      }
      std::string err;
      const bool ok = read_stream(paths[(size_t)j.item * n_streams + j.stream].c_str(), j.stream, j.start, T, augment, normalize,
                                  j.slot->wave + ((size_t)j.b * n_streams + j.stream) * T,
                                  j.slot->len + (size_t)j.b * n_streams + j.stream, j.slot->stat + 2 * (size_t)j.b, err);
      {
        std::lock_guard<std::mutex> lk(mu);
        if (!ok && j.slot->err.empty()) j.slot->err = err;
        if (--j.slot->pending == 0) cv_done.notify_all();
      }
    }
  }
};

extern "C" int srf_feeder_create(const char* const* paths, int n_items, int n_streams, int time_samples, int batch,
                                 int n_threads, int augment, int shuffle, int drop_last, unsigned long long seed,
                                 srf_feeder** out) {
  return srf_feeder_create_sharded(paths, n_items, n_streams, time_samples, batch, n_threads, augment, shuffle, drop_last, seed,
                                   /*normalize=*/1, /*rank=*/0, /*world=*/1, out);
}

// Rank-aware form (one process per GPU; the reference feeds all its DataParallel replicas from ONE DataLoader,
// wham.py:219-226, which then scatters every batch): all ranks build the same epoch order from (seed, epoch), a GLOBAL batch
// is `batch * world` consecutive items of it, and this feeder delivers items [rank * batch, (rank + 1) * batch) of every
// global batch -- so the ranks' batches of a step are disjoint, their concatenation in rank order IS the single-process
// batch of size batch * world, and an epoch covers every item exactly once over all ranks.  world > 1 needs drop_last
// (every rank must see the same number of steps, or the gradient all-reduce deadlocks).
extern "C" int srf_feeder_create_sharded(const char* const* paths, int n_items, int n_streams, int time_samples, int batch,
                                         int n_threads, int augment, int shuffle, int drop_last, unsigned long long seed,
                                         int normalize, int rank, int world, srf_feeder** out) {
  SRF_CHECK_ARG(paths && out && n_items > 0 && n_streams > 0 && time_samples > 0 && batch > 0 && n_threads > 0,
                "srf_feeder_create: bad arguments");
  SRF_CHECK_ARG(world >= 1 && rank >= 0 && rank < world, "srf_feeder_create: rank %d outside world %d", rank, world);
  SRF_CHECK_ARG(world == 1 || drop_last, "srf_feeder_create: a sharded feeder (world %d) needs drop_last", world);
  *out = nullptr;
  srf_feeder* f = new srf_feeder();
  f->normalize = normalize;
  f->rank = rank;
  f->world = world;
  f->n_items = n_items;
  f->n_streams = n_streams;
  f->T = time_samples;
  f->batch = batch;
  f->augment = augment;
  f->shuffle = shuffle;
  f->drop_last = drop_last;
  f->seed = seed;
  f->paths.reserve((size_t)n_items * n_streams);
  for (long i = 0; i < (long)n_items * n_streams; ++i) {
    if (!paths[i]) {
      delete f;
      SRF_CHECK_ARG(false, "srf_feeder_create: null path %ld", i);
    }
    f->paths.emplace_back(paths[i]);
  }
  f->frames.resize(n_items);
  for (int i = 0; i < n_items; ++i) {   // the reference's metadata pass (wham.py:111-120): mixture lengths
    int ch = 0;
    long fr = 0;
    if (srf_wav_info(f->paths[(size_t)i * n_streams].c_str(), nullptr, &ch, nullptr, &fr) != SRF_OK) {
      delete f;
      return SRF_EINVAL;   // srf_last_error is set
    }
    f->frames[i] = fr;
  }
  f->order.resize(n_items);
  for (int i = 0; i < n_items; ++i) f->order[i] = i;
  for (int t = 0; t < n_threads; ++t) f->workers.emplace_back([f] { f->work(); });
  *out = f;
  return SRF_OK;
}

extern "C" void srf_feeder_destroy(srf_feeder* f) {
  if (!f) return;
  {
    std::lock_guard<std::mutex> lk(f->mu);
    f->stop = true;
    f->jobs.clear();
  }
  f->cv_job.notify_all();
  for (auto& t : f->workers) t.join();
  for (auto* s : f->inflight) delete s;
  delete f;
}

extern "C" long srf_feeder_batches_per_epoch(const srf_feeder* f) {
  if (!f) return 0;
  const long g = (long)f->batch * f->world;
  return f->drop_last ? f->n_items / g : (f->n_items + g - 1) / g;
}

// The items this rank delivers in the CURRENT epoch, in delivery order (after srf_feeder_start_epoch): fills at most
// `capacity` entries, returns the count.  For bookkeeping and for the tests of the sharding.
extern "C" long srf_feeder_epoch_items(srf_feeder* f, int* items, long capacity) {
  if (!f) return 0;
  std::lock_guard<std::mutex> lk(f->mu);
  const long g = (long)f->batch * f->world;
  long n = 0;
  for (long base = 0; base < f->n_items; base += g) {
    const long left = f->n_items - base;
    if (f->drop_last && left < g) break;
    for (long b = 0; b < f->batch; ++b) {
      const long pos = base + (long)f->rank * f->batch + b;
      if (pos >= f->n_items) break;
      if (items && n < capacity) items[n] = f->order[pos];
      ++n;
    }
  }
  return n;
}

extern "C" long srf_feeder_item_frames(const srf_feeder* f, int item) {
  return (f && item >= 0 && item < f->n_items) ? f->frames[item] : -1;
}

// New epoch: item order (Fisher-Yates from splitmix64(seed, epoch) when shuffling) and cursor reset.  Outstanding slots
// must have been waited for.
extern "C" int srf_feeder_start_epoch(srf_feeder* f, int epoch) {
  SRF_CHECK_ARG(f, "srf_feeder_start_epoch: null feeder");
  std::lock_guard<std::mutex> lk(f->mu);
  SRF_CHECK_ARG(f->inflight.empty(), "srf_feeder_start_epoch: %zu submitted batches have not been waited for", f->inflight.size());
  f->epoch = epoch;
  f->cursor = 0;
  for (int i = 0; i < f->n_items; ++i) f->order[i] = i;
  if (f->shuffle) {
    uint64_t s = splitmix64(f->seed ^ (0x5851F42D4C957F2DULL * (uint64_t)(epoch + 1)));
    for (int i = f->n_items - 1; i > 0; --i) {
      s = splitmix64(s);
      const int j = (int)(s % (uint64_t)(i + 1));
      std::swap(f->order[i], f->order[j]);
    }
  }
  return SRF_OK;
}

// Queue the next batch of the epoch into caller-owned buffers (wave: batch * streams * time floats, len: batch * streams ints; pinned
// host memory if the caller wants an asynchronous copy).  Returns 1 when the epoch has no batch left (nothing queued).
extern "C" int srf_feeder_submit(srf_feeder* f, float* wave, int* len, float* stat) {
  SRF_CHECK_ARG(f && wave && len && stat, "srf_feeder_submit: null pointer");
  std::lock_guard<std::mutex> lk(f->mu);
  const long gbatch = (long)f->batch * f->world;
  const long left = f->n_items - f->cursor;                       // items of the epoch not yet dealt to a global batch
  if (left <= 0 || (f->drop_last && left < gbatch)) return 1;
  const long mine0 = f->cursor + (long)f->rank * f->batch;        // this rank's slice of the global batch
  const long avail = f->n_items - mine0;
  const int nb = avail < f->batch ? (int)(avail > 0 ? avail : 0) : f->batch;   // (< batch only without drop_last, world 1)
  auto* s = new srf_feeder::Slot();
  s->wave = wave;
  s->len = len;
  s->stat = stat;
  s->n_valid = nb;
  s->pending = nb * f->n_streams;
  for (int b = 0; b < f->batch; ++b) {
    for (int st = 0; st < f->n_streams; ++st) len[(size_t)b * f->n_streams + st] = 0;
    stat[2 * b] = 0.f;
    stat[2 * b + 1] = 1.f;
  }
  if (nb < f->batch) memset(wave + (size_t)nb * f->n_streams * f->T, 0, sizeof(float) * (size_t)(f->batch - nb) * f->n_streams * f->T);
  for (int b = 0; b < nb; ++b) {
    const int item = f->order[mine0 + b];
    long start = 0;
    if (f->augment && f->frames[item] > f->T) {   // wham.py:183-186: one start per example, shared by its files
      const uint64_t r = splitmix64(splitmix64(f->seed + 0x632BE59BD9B4E019ULL * (uint64_t)(f->epoch + 1)) ^ (uint64_t)item);
      start = (long)(r % (uint64_t)(f->frames[item] - f->T));
    }
    for (int st = 0; st < f->n_streams; ++st) f->jobs.push_back(srf_feeder::Job{s, b, item, st, start});
  }
  f->cursor += gbatch;
  f->inflight.push_back(s);
  f->cv_job.notify_all();
  return SRF_OK;
}

// Block until the oldest submitted batch is complete; hands back its buffers and the number of real examples in it.
extern "C" int srf_feeder_wait(srf_feeder* f, float** wave, int** len, float** stat, int* n_valid) {
  SRF_CHECK_ARG(f, "srf_feeder_wait: null feeder");
  srf_feeder::Slot* s = nullptr;
  {
    std::unique_lock<std::mutex> lk(f->mu);
    SRF_CHECK_ARG(!f->inflight.empty(), "srf_feeder_wait: nothing submitted");
    s = f->inflight.front();
    f->cv_done.wait(lk, [&] { return s->pending.load() == 0; });
    f->inflight.pop_front();
  }
  if (wave) *wave = s->wave;
  if (len) *len = s->len;
  if (stat) *stat = s->stat;
  if (n_valid) *n_valid = s->n_valid;
  const std::string err = s->err;
  delete s;
  SRF_CHECK_ARG(err.empty(), "srf_feeder: %s", err.c_str());
  return SRF_OK;
}

// One example, synchronously in the calling thread (Dataset.__getitem__'s reads): paths[n_streams] = mixture, sources; the
// crop [start, start + time_samples) of every file -> wave [n_streams][time_samples], len [n_streams], stat [2] exactly as
// a worker thread fills them for one batch row.
extern "C" int srf_feeder_read_example(const char* const* paths, int n_streams, int time_samples, long start, int augment,
                                       int normalize, float* wave, int* len, float* stat) {
  SRF_CHECK_ARG(paths && wave && len && stat && n_streams > 0 && time_samples > 0 && start >= 0,
                "srf_feeder_read_example: bad arguments");
  stat[0] = 0.f;
  stat[1] = 1.f;
  for (int st = 0; st < n_streams; ++st) {
    SRF_CHECK_ARG(paths[st], "srf_feeder_read_example: null path %d", st);
    std::string err;
    const bool ok = read_stream(paths[st], st, start, time_samples, augment, normalize, wave + (size_t)st * time_samples,
                                len + st, stat, err);
    SRF_CHECK_ARG(ok, "srf_feeder: %s", err.c_str());
  }
  return SRF_OK;
}

// ---------------------------------------------------------------------------------------------
// device side: the Dataset's normalisation recipe on a whole batch
// ---------------------------------------------------------------------------------------------
// raw [B][S1][T] (stream 0 = mixture; every stream zero padded beyond its len[b][s]) -> mix [B][T], src [B][S1-1][T].
// normalize = 0: plain copy (wham.py:190-191,208-209 skipped).  normalize = 1, wham.py:189-217:
//   every stream: x <- (x - mean) / (std + eps) over its VALID samples (torch .std(): unbiased), then zero pad.  The
//   mixture's {mean, std} come from the host (stat[b]): without the random crop the reference normalises the mixture over
//   the WHOLE file before it truncates to T (wham.py:183-191), while the sources are sliced first (:201);
//   mix_std = population std of the padded mixture (numpy .std());  every stream: x <- (x - mean_T(x)) / (mix_std + eps)
//   with mean_T over all T samples of the padded signal.
// One block per example; sums in fp64.
__device__ __forceinline__ double srf_feeder_block_sum(double v, double* red) {   // every thread gets the total
  v = srf_wave_sum(v);
  __syncthreads();   // (red may still be read from the previous call)
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void srf_feeder_normalize_kernel(const float* __restrict__ raw, const int* __restrict__ len,
                                                                   const float* __restrict__ stat, int S1, int T,
                                                                   int normalize, float eps, float* __restrict__ mix,
                                                                   float* __restrict__ src) {
  __shared__ double red[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  double mix_std = 0.0;
  for (int s = 0; s < S1; ++s) {
    // valid samples of THIS stream: the reference normalises the slice it read (wham.py:201-207), so a source file shorter
    // than its mixture must not have its zero padding counted in its mean / std
    const int n = min(max(len[(size_t)b * S1 + s], 0), T);
    const float* x = raw + ((size_t)b * S1 + s) * T;
    float* y = s == 0 ? mix + (size_t)b * T : src + ((size_t)b * (S1 - 1) + (s - 1)) * T;
    if (!normalize) {
      for (int i = tid; i < T; i += 256) y[i] = i < n ? x[i] : 0.f;
      continue;
    }
    float m32, den;
    if (s == 0) {
      m32 = stat[2 * b];
      den = stat[2 * b + 1] + eps;
    } else {
      double a = 0.0, q = 0.0;
      for (int i = tid; i < n; i += 256) a += (double)x[i];
      const double mean = n > 0 ? srf_feeder_block_sum(a, red) / n : 0.0;
      for (int i = tid; i < n; i += 256) {
        const double d = (double)x[i] - mean;
        q += d * d;
      }
      const double ss = srf_feeder_block_sum(q, red);
      const double var = n > 1 ? ss / (n - 1) : 0.0 / 0.0;   // torch.std of one sample is NaN
      m32 = (float)mean;
      den = (float)sqrt(var) + eps;
    }
    // the reference's fp32 arithmetic: (x - mean) / (std + eps)
    double a2 = 0.0;
    for (int i = tid; i < T; i += 256) {
      const float v = i < n ? (x[i] - m32) / den : 0.f;
      y[i] = v;
      a2 += (double)v;
    }
    const double mean2 = srf_feeder_block_sum(a2, red) / T;
    if (s == 0) {   // numpy population std of the padded, normalised mixture
      double q2 = 0.0;
      for (int i = tid; i < T; i += 256) {
        const double d = (double)y[i] - mean2;
        q2 += d * d;
      }
      mix_std = sqrt(srf_feeder_block_sum(q2, red) / T);
    }
    const float m2 = (float)mean2, den2 = (float)mix_std + eps;
    for (int i = tid; i < T; i += 256) y[i] = (y[i] - m2) / den2;
  }
}

extern "C" int srf_feeder_normalize(const float* raw, const int* len, const float* stat, int B, int n_streams, int T,
                                    int normalize, float eps, float* mix, float* src, void* stream) {
  SRF_CHECK_ARG(raw && len && mix && (src || n_streams == 1) && (stat || !normalize), "srf_feeder_normalize: null pointer");
  SRF_CHECK_ARG(B > 0 && n_streams > 0 && T > 0, "srf_feeder_normalize: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(srf_feeder_normalize_kernel, dim3((unsigned)B), dim3(256), 0, st, raw, len, stat, n_streams, T, normalize,
                     eps, mix, src);
  SRF_CHECK_LAUNCH("feeder_normalize", st);
  return SRF_OK;
}
