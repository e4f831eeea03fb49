// mailbox.hip -- one-shot peer-to-peer exchange of small records between the ranks of one node (SURVEY 5 "comm backend",
// 8e: "a one-shot P2P mailbox exchange ... fixed rank-order summation -> identical bits on every rank").
//
// What it replaces: a collective of a few hundred bytes issued from the host through torch.distributed / RCCL -- 19 per
// frame for the ICP normal equations in the row-band scheme (the reference's analogue is the 168-byte device-to-host copy
// per iteration, src/sensor/localization_kernels.cu:318-325), one per chunk for the 80-byte pose records of the
// frame-sharded scheme.  At this size a collective is pure latency (~25 us each through RCCL between 8 ranks, plus the
// host round trip that issues it); here a rank STORES its record straight into every peer's inbox over xGMI and polls its
// own -- two short launches on the caller's stream, no host involvement, no library call.
//
// Layout: every rank owns an inbox [kSlots][world][kMaxGranules] of 8-byte granules {tag = epoch, value = 4 payload
// bytes} in device memory that its peers have mapped (hipIpcOpenMemHandle across processes; plain pointers inside one
// process).  The granule is the hand-off of cdna_hip_programming.md Guideline 16, form R2 -- the data is its own flag, one aligned
// 8-byte store per granule -- widened from agent to SYSTEM scope, since writer and reader are different devices.
// Epochs count the collectives of a mailbox (every rank calls them in the same order); the inbox ring needs two slots:
// a rank can only write epoch e + 2 after it has read epoch e + 1 from every peer, which every peer wrote after it had
// finished reading epoch e.  Sums are formed in rank order from the gathered records: the same bits on every rank (and,
// the ICP sums being exact integers, the bits of the one-GPU run).
#include <string.h>

#include <vector>

#include "common.hpp"

namespace svoslam {

constexpr int kMbSlots = 4;
constexpr int kMbMaxGranules = 512;      // 2 KB of payload per rank and collective
constexpr unsigned kMbSpinLimit = 1u << 24;  // default polls per granule before a wait gives up (svoslam_mailbox_set_wait_limit)

__global__ void mailbox_post_kernel(unsigned long long *const *__restrict__ inboxes, int world, int rank, int slot, unsigned epoch,
                                    const unsigned *__restrict__ src, int granules) {
  // thread t, peer p: granule t of this rank's record -> inbox of p
  for (int p = (int)blockIdx.x; p < world; p += (int)gridDim.x) {
    unsigned long long *dst = inboxes[p] + ((size_t)slot * world + rank) * kMbMaxGranules;
    for (int g = (int)threadIdx.x; g < granules; g += (int)blockDim.x)
      __hip_atomic_store(dst + g, ((unsigned long long)epoch << 32) | src[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// waits until every rank's record of this epoch has arrived in the OWN inbox, then copies the payloads out in rank order;
// reduce_f64 != 0: out[k] = sum over ranks (in rank order) of the k-th double instead
__global__ void mailbox_collect_kernel(const unsigned long long *__restrict__ inbox, int world, int slot, unsigned epoch,
                                       unsigned *__restrict__ out, int granules, int reduce_f64, unsigned *__restrict__ fail,
                                       unsigned spin_limit) {
  __shared__ unsigned vals[kMbMaxGranules];
  __shared__ int bad;
  if (threadIdx.x == 0) bad = 0;
  __syncthreads();
  double acc = 0.0;  // thread k < granules / 2 owns the k-th double (reduce)
  for (int r = 0; r < world; r++) {
    const unsigned long long *src = inbox + ((size_t)slot * world + r) * kMbMaxGranules;
    for (int g = (int)threadIdx.x; g < granules; g += (int)blockDim.x) {
      unsigned long long x = 0;
      unsigned spins = 0;
      for (;;) {
        x = __hip_atomic_load(src + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((unsigned)(x >> 32) == epoch) break;
        // a peer that never posted (about a second of polling): the wait gives up LOUDLY -- the granule is poisoned (all ones: a NaN
        // as binary32 and as either half of a binary64, so a pose record or an ICP sum built from it cannot pass for data) and
        // the mailbox's sticky fail word is set, which DistContext checks at the end of every stream call
        if (++spins > spin_limit) { bad = 1; x = 0xFFFFFFFFull; break; }
        __builtin_amdgcn_s_sleep(2);
      }
      if (reduce_f64) vals[g] = (unsigned)x;
      else out[(size_t)r * granules + g] = (unsigned)x;
    }
    if (reduce_f64) {
      __syncthreads();
      const int k = (int)threadIdx.x;
      if (2 * k + 1 < granules) {
        const unsigned long long bits = ((unsigned long long)vals[2 * k + 1] << 32) | vals[2 * k];
        acc += __longlong_as_double((long long)bits);
      }
      __syncthreads();
    }
  }
  if (reduce_f64) {
    const int k = (int)threadIdx.x;
    if (2 * k + 1 < granules) reinterpret_cast<double *>(out)[k] = acc;
  }
  __syncthreads();
  if (threadIdx.x == 0 && bad && fail) *fail = 1u;
}

}  // namespace svoslam

struct svoslam_mailbox {
  int rank = 0, world = 1;
  unsigned long long *inbox = nullptr;          // own
  std::vector<unsigned long long *> peers;      // [world], own entry = inbox
  std::vector<bool> opened;                     // peers mapped through IPC handles (to be closed)
  unsigned long long **d_peers = nullptr;       // device copy of the pointer table
  unsigned *d_stage = nullptr;                  // reduce: staging of the local record (so that src may alias dst)
  unsigned *d_fail = nullptr;
  unsigned epoch = 0;
  unsigned spin_limit = svoslam::kMbSpinLimit;
  bool connected = false;
};

using namespace svoslam;

extern "C" {

int svoslam_mailbox_create(svoslam_mailbox **out, int32_t rank, int32_t world) {
  if (!out || world < 1 || rank < 0 || rank >= world) return SVOSLAM_ERR_INVALID_ARG;
  SVO_TRY(ensure_device());
  svoslam_mailbox *m = new svoslam_mailbox();
  m->rank = rank; m->world = world;
  const size_t bytes = (size_t)kMbSlots * world * kMbMaxGranules * 8;
  // fine-grained device memory where the runtime offers it (stores of another device become visible without a kernel
  // boundary); plain device memory otherwise (same-device peers: the tests' arrangement)
  if (hipExtMallocWithFlags((void **)&m->inbox, bytes, hipDeviceMallocFinegrained) != hipSuccess) {
    (void)hipGetLastError();
    if (hipMalloc((void **)&m->inbox, bytes) != hipSuccess) { delete m; return SVOSLAM_ERR_OOM; }
  }
  m->peers.assign((size_t)world, nullptr);
  m->opened.assign((size_t)world, false);
  // (a failure below releases what exists so far: svoslam_mailbox_destroy takes the half-built object)
  if (memset_sync(m->inbox, 0, bytes) != hipSuccess ||                                  // tag 0 is never an epoch
      hipMalloc((void **)&m->d_peers, (size_t)world * sizeof(void *)) != hipSuccess ||
      hipMalloc((void **)&m->d_stage, kMbMaxGranules * 4) != hipSuccess || hipMalloc((void **)&m->d_fail, 4) != hipSuccess ||
      memset_sync(m->d_fail, 0, 4) != hipSuccess) {
    (void)hipGetLastError();
    (void)svoslam_mailbox_destroy(m);
    return SVOSLAM_ERR_HIP;
  }
  m->peers[(size_t)rank] = m->inbox;
  *out = m;
  return SVOSLAM_OK;
}

int svoslam_mailbox_destroy(svoslam_mailbox *m) {
  if (!m) return SVOSLAM_OK;
  (void)hipDeviceSynchronize();
  for (int p = 0; p < m->world; p++)
    if (m->opened[(size_t)p] && m->peers[(size_t)p]) (void)hipIpcCloseMemHandle(m->peers[(size_t)p]);
  if (m->inbox) (void)hipFree(m->inbox);
  if (m->d_peers) (void)hipFree(m->d_peers);
  if (m->d_stage) (void)hipFree(m->d_stage);
  if (m->d_fail) (void)hipFree(m->d_fail);
  delete m;
  return SVOSLAM_OK;
}

// 64 bytes that another PROCESS turns into a mapping of this rank's inbox (exchange them with any host-side all-gather)
int svoslam_mailbox_handle(svoslam_mailbox *m, void *handle64) {
  if (!m || !handle64) return SVOSLAM_ERR_INVALID_ARG;
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "handle size");
  hipIpcMemHandle_t h;
  SVO_HIP(hipIpcGetMemHandle(&h, m->inbox));
  memcpy(handle64, &h, 64);
  return SVOSLAM_OK;
}

static int finish_connect(svoslam_mailbox *m) {
  for (int p = 0; p < m->world; p++)
    if (!m->peers[(size_t)p]) return SVOSLAM_ERR_INVALID_ARG;
  SVO_HIP(hipMemcpy(m->d_peers, m->peers.data(), (size_t)m->world * sizeof(void *), hipMemcpyHostToDevice));
  m->connected = true;
  return SVOSLAM_OK;
}

// handles = world x 64 bytes (entry `rank` is ignored)
int svoslam_mailbox_connect(svoslam_mailbox *m, const void *handles) {
  if (!m || !handles) return SVOSLAM_ERR_INVALID_ARG;
  for (int p = 0; p < m->world; p++) {
    if (p == m->rank) continue;
    hipIpcMemHandle_t h;
    memcpy(&h, (const char *)handles + 64 * (size_t)p, 64);
    void *ptr = nullptr;
    if (hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
      (void)hipGetLastError();
      for (int q = 0; q < m->world; q++)  // no half-registered mappings: close what this call opened
        if (m->opened[(size_t)q] && m->peers[(size_t)q]) { (void)hipIpcCloseMemHandle(m->peers[(size_t)q]); m->peers[(size_t)q] = nullptr; m->opened[(size_t)q] = false; }
      return SVOSLAM_ERR_HIP;
    }
    m->peers[(size_t)p] = (unsigned long long *)ptr;
    m->opened[(size_t)p] = true;
  }
  return finish_connect(m);
}

// peers inside ONE process (several devices with peer access enabled, or several mailboxes on one device: the tests)
int svoslam_mailbox_connect_local(svoslam_mailbox *m, svoslam_mailbox *const *all) {
  if (!m || !all) return SVOSLAM_ERR_INVALID_ARG;
  for (int p = 0; p < m->world; p++) {
    if (!all[p] || all[p]->world != m->world || all[p]->rank != p) return SVOSLAM_ERR_INVALID_ARG;
    m->peers[(size_t)p] = all[p]->inbox;
  }
  return finish_connect(m);
}

// the two halves of a collective.  post: this rank's record goes to every inbox (epoch advanced).  collect: wait for every
// rank's record of the current epoch, copy out / sum.  A caller that drives SEVERAL mailboxes from one stream (tests)
// posts them all before it collects any: a collect kernel occupies its stream until its peers have posted.
static int post(svoslam_mailbox *m, const void *d_src, int bytes, bool stage, hipStream_t s) {
  if (!m || !m->connected || !d_src || bytes <= 0 || (bytes & 3) || bytes > kMbMaxGranules * 4) return SVOSLAM_ERR_INVALID_ARG;
  m->epoch += 1;
  if (m->epoch == 0) m->epoch = 1;   // tag 0 = "nothing yet"
  const int slot = (int)(m->epoch % kMbSlots);
  const unsigned *src = reinterpret_cast<const unsigned *>(d_src);
  if (stage) {  // the record may be reduced in place: post from a private copy
    SVO_HIP(hipMemcpyAsync(m->d_stage, d_src, (size_t)bytes, hipMemcpyDeviceToDevice, s));
    src = m->d_stage;
  }
  mailbox_post_kernel<<<m->world, 64, 0, s>>>(m->d_peers, m->world, m->rank, slot, m->epoch, src, bytes / 4);
  SVO_LAUNCH_CHECK();
  return SVOSLAM_OK;
}

static int collect(svoslam_mailbox *m, void *d_dst, int bytes, int reduce, hipStream_t s) {
  if (!m || !m->connected || !d_dst || bytes <= 0 || (bytes & 3) || bytes > kMbMaxGranules * 4 || m->epoch == 0) return SVOSLAM_ERR_INVALID_ARG;
  if (reduce && (bytes & 7)) return SVOSLAM_ERR_INVALID_ARG;
  const int slot = (int)(m->epoch % kMbSlots);
  mailbox_collect_kernel<<<1, 256, 0, s>>>(m->inbox, m->world, slot, m->epoch, reinterpret_cast<unsigned *>(d_dst), bytes / 4, reduce, m->d_fail, m->spin_limit);
  SVO_LAUNCH_CHECK();
  return SVOSLAM_OK;
}

int svoslam_mailbox_post(svoslam_mailbox *m, const void *d_src, int32_t bytes, void *stream) {
  return post(m, d_src, bytes, true, reinterpret_cast<hipStream_t>(stream));
}
int svoslam_mailbox_collect(svoslam_mailbox *m, void *d_dst, int32_t bytes, int32_t reduce_f64, void *stream) {
  return collect(m, d_dst, bytes, reduce_f64 != 0, reinterpret_cast<hipStream_t>(stream));
}

// d_dst[world][bytes] <- every rank's d_src[bytes] (bytes a multiple of 4, <= 2048); enqueued on `stream`
int svoslam_mailbox_all_gather(svoslam_mailbox *m, const void *d_src, int32_t bytes, void *d_dst, void *stream) {
  if (!d_dst) return SVOSLAM_ERR_INVALID_ARG;
  SVO_TRY(post(m, d_src, bytes, false, reinterpret_cast<hipStream_t>(stream)));
  return collect(m, d_dst, bytes, 0, reinterpret_cast<hipStream_t>(stream));
}

// d_values[count] <- sum over ranks, added in rank order (count <= 256 doubles); in place; enqueued on `stream`
int svoslam_mailbox_all_reduce_f64(svoslam_mailbox *m, double *d_values, int32_t count, void *stream) {
  if (count <= 0 || count > kMbMaxGranules / 2) return SVOSLAM_ERR_INVALID_ARG;
  SVO_TRY(post(m, d_values, count * 8, true, reinterpret_cast<hipStream_t>(stream)));
  return collect(m, d_values, count * 8, 1, reinterpret_cast<hipStream_t>(stream));
}

// polls per granule before a wait gives up (default 2^24, about a second); tests shorten it
int svoslam_mailbox_set_wait_limit(svoslam_mailbox *m, uint32_t polls) {
  if (!m || polls == 0) return SVOSLAM_ERR_INVALID_ARG;
  m->spin_limit = polls;
  return SVOSLAM_OK;
}

// 1 if a wait of this mailbox has ever given up (a peer that never posted).  Blocking.
int svoslam_mailbox_failed(svoslam_mailbox *m, int32_t *failed) {
  if (!m || !failed) return SVOSLAM_ERR_INVALID_ARG;
  unsigned f = 0;
  SVO_HIP(hipDeviceSynchronize());
  SVO_HIP(hipMemcpy(&f, m->d_fail, 4, hipMemcpyDeviceToHost));
  *failed = f ? 1 : 0;
  return SVOSLAM_OK;
}

}  // extern "C"
