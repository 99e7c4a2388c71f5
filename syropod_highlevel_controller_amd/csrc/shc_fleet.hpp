// shc_fleet.hpp — shc_fleet_*: mixed-morphology batches and multi-device sharding behind one handle (C ABI, host side).
//
// BASELINE.json configs[3] / configs[4]: robots are independent (nothing in the reference couples two robots), so a batch
// shards by contiguous instance ranges with no data-path exchange while stepping, and a mixed-morphology batch is binned by
// morphology: the cycle kernel keeps ONE morphology's DH / limit tables in LDS and maps one leg to one lane, so every
// (morphology bin, device) pair gets its own engine on its own HIP stream.  The fleet keeps the caller's instance order at
// the boundary whatever the interleaving pattern.  The only exchange is the all-gather of the final joint-state buffer:
// every device ends up with every instance's joints, copied device to device (hipMemcpyPeerAsync: xGMI on an MI355X node).
// One-process-per-GPU hosts (bench.py under torch.distributed) do the same exchange with RCCL instead.
#pragma once

#include <algorithm>
#include <cmath>
#include <vector>

struct FleetPart {
  shc_engine *engine = nullptr;
  int morph = 0, device = 0, device_slot = 0;
  hipStream_t stream = nullptr;
  std::vector<int64_t> ids; // caller's instance ids, ascending
  // the exchange step's buffers, allocated by the first shc_fleet_all_gather_joints and kept
  double *g_joints = nullptr;         // this part's joints [rows][L][D] on its own device
  int64_t *g_ids = nullptr;           // its ids on its own device
  hipEvent_t g_ready = nullptr;       // recorded on `stream` once g_joints is complete
  std::vector<double *> r_joints;     // per device slot (other devices only): the landing buffer of the peer copy ...
  std::vector<int64_t *> r_ids;       // ... and the ids over there
};

struct shc_fleet {
  std::vector<shc_params> params;
  std::vector<FleetPart> parts;
  std::vector<int> devices;
  int64_t n = 0;
  int max_legs = 0, max_dof = 0;
  std::vector<double *> gather; // per device: [n][max_legs][max_dof], NaN padded
  std::vector<hipStream_t> gather_stream; // per device: incoming peer copies + their placement
  std::vector<double> host_a, host_b;
  std::vector<int32_t> host_i;
};

static void fleet_free(shc_fleet *f) {
  if (!f) return;
  for (auto &p : f->parts) {
    if (p.engine) shc_engine_destroy(p.engine);
    if (p.stream) {
      (void)hipSetDevice(p.device);
      (void)hipStreamDestroy(p.stream);
    }
  }
  for (auto &p : f->parts) {
    (void)hipSetDevice(p.device);
    (void)hipFree(p.g_joints);
    (void)hipFree(p.g_ids);
    if (p.g_ready) (void)hipEventDestroy(p.g_ready);
    for (size_t d = 0; d < p.r_joints.size(); ++d) {
      (void)hipSetDevice(f->devices[d]);
      (void)hipFree(p.r_joints[d]);
      (void)hipFree(p.r_ids[d]);
    }
  }
  for (size_t d = 0; d < f->gather.size(); ++d) {
    (void)hipSetDevice(f->devices[d]);
    if (f->gather[d]) (void)hipFree(f->gather[d]);
    if (d < f->gather_stream.size() && f->gather_stream[d]) (void)hipStreamDestroy(f->gather_stream[d]);
  }
  delete f;
}

static int f_parts_on_device(const shc_fleet *f, int device) {
  int k = 0;
  for (const auto &p : f->parts) k += p.device == device ? 1 : 0;
  return k;
}
extern "C" int shc_fleet_create(const shc_params *params, int n_morphologies, const int32_t *morph_id, int64_t n_instances, const int *device_ids,
                                int n_devices, shc_fleet **out) {
  if (!params || !out || n_morphologies < 1 || n_instances < 1) return fail(SHC_ERR_INVALID_ARG, "params / out NULL, or no morphology / instance");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(SHC_ERR_NO_DEVICE, "no HIP device visible: the engine has no CPU fallback");
  const int one = 0;
  if (!device_ids || n_devices < 1) {
    device_ids = &one;
    n_devices = 1;
  }
  for (int d = 0; d < n_devices; ++d)
    if (device_ids[d] < 0 || device_ids[d] >= ndev) return fail(SHC_ERR_INVALID_ARG, "device id out of range");
  shc_fleet *f = new shc_fleet();
  f->params.assign(params, params + n_morphologies);
  f->devices.assign(device_ids, device_ids + n_devices);
  f->n = n_instances;
  std::vector<std::vector<int64_t>> bins(n_morphologies);
  for (int64_t i = 0; i < n_instances; ++i) {
    const int m = morph_id ? morph_id[i] : 0;
    if (m < 0 || m >= n_morphologies) {
      fleet_free(f);
      return fail(SHC_ERR_INVALID_ARG, "morph_id out of range");
    }
    bins[m].push_back(i);
  }
  for (int m = 0; m < n_morphologies; ++m) {
    if (bins[m].empty()) continue;
    int L, NJ;
    const int rc = validate_params(&params[m], &L, &NJ);
    if (rc != SHC_OK) {
      fleet_free(f);
      return rc;
    }
    f->max_legs = std::max(f->max_legs, L);
    f->max_dof = std::max(f->max_dof, NJ);
  }
  // one init chain per morphology, shared by its shards
  for (int m = 0; m < n_morphologies; ++m) {
    if (bins[m].empty()) continue;
    shc_tables tables;
    int rc = shc_generate_tables(&params[m], &tables);
    const int64_t nb = int64_t(bins[m].size());
    for (int d = 0; d < n_devices && rc == SHC_OK; ++d) { // contiguous shards of the bin, sizes differing by at most one
      const int64_t base = nb / n_devices, rem = nb % n_devices;
      const int64_t lo = d * base + std::min<int64_t>(d, rem), hi = lo + base + (d < rem ? 1 : 0);
      if (hi == lo) continue;
      FleetPart part;
      part.morph = m;
      part.device = device_ids[d];
      part.device_slot = d;
      part.ids.assign(bins[m].begin() + lo, bins[m].begin() + hi);
      if (hipSetDevice(part.device) != hipSuccess || hipStreamCreateWithFlags(&part.stream, hipStreamNonBlocking) != hipSuccess) {
        rc = fail(SHC_ERR_HIP, "stream creation failed");
        break;
      }
      rc = shc_engine_create_with_tables(&params[m], &tables, hi - lo, part.device, part.stream, &part.engine);
      f->parts.push_back(part); // (also on failure: fleet_free releases the stream)
    }
    if (rc != SHC_OK) {
      fleet_free(f);
      return rc;
    }
  }
  // parts that share a device already run concurrently, one stream each: no two-stream split inside such a part (shc_engine_step)
  for (auto &part : f->parts)
    if (f_parts_on_device(f, part.device) > 1) {
      const int rc = shc_engine_set_features(part.engine, part.engine->features | SHC_FEAT_SINGLE_STREAM);
      if (rc != SHC_OK) {
        fleet_free(f);
        return rc;
      }
    }
  f->gather.assign(n_devices, nullptr);
  *out = f;
  return SHC_OK;
}

extern "C" int shc_fleet_destroy(shc_fleet *f) {
  fleet_free(f);
  return SHC_OK;
}
extern "C" int64_t shc_fleet_instances(const shc_fleet *f) { return f ? f->n : 0; }
extern "C" int shc_fleet_part_count(const shc_fleet *f) { return f ? int(f->parts.size()) : 0; }
extern "C" int shc_fleet_part(const shc_fleet *f, int k, shc_engine **engine, int *morphology, int *device, int64_t *n_instances) {
  if (!f || k < 0 || k >= int(f->parts.size())) return fail(SHC_ERR_INVALID_ARG, "part index out of range");
  const FleetPart &p = f->parts[k];
  if (engine) *engine = p.engine;
  if (morphology) *morphology = p.morph;
  if (device) *device = p.device;
  if (n_instances) *n_instances = int64_t(p.ids.size());
  return SHC_OK;
}
extern "C" int shc_fleet_part_instances(const shc_fleet *f, int k, int64_t *ids) {
  if (!f || !ids || k < 0 || k >= int(f->parts.size())) return fail(SHC_ERR_INVALID_ARG, "part index out of range / ids NULL");
  std::copy(f->parts[k].ids.begin(), f->parts[k].ids.end(), ids);
  return SHC_OK;
}

// caller-order rows of `width` doubles -> the part's rows
static const double *fleet_pick(shc_fleet *f, const FleetPart &p, const double *src, int width, std::vector<double> &buf) {
  if (!src) return nullptr;
  buf.resize(p.ids.size() * size_t(width));
  for (size_t k = 0; k < p.ids.size(); ++k) std::copy(src + p.ids[k] * width, src + (p.ids[k] + 1) * width, buf.begin() + k * width);
  return buf.data();
}

extern "C" int shc_fleet_set_velocity(shc_fleet *f, const double *linear_xy, const double *angular) {
  if (!f) return fail(SHC_ERR_INVALID_ARG, "fleet is NULL");
  for (auto &p : f->parts) {
    const int rc = shc_engine_set_velocity(p.engine, fleet_pick(f, p, linear_xy, 2, f->host_a), fleet_pick(f, p, angular, 1, f->host_b), 0);
    if (rc != SHC_OK) return rc;
  }
  return SHC_OK;
}
extern "C" int shc_fleet_set_imu(shc_fleet *f, const double *orientation_wxyz, const double *angular_velocity) {
  if (!f) return fail(SHC_ERR_INVALID_ARG, "fleet is NULL");
  for (auto &p : f->parts) {
    const int rc = shc_engine_set_imu(p.engine, fleet_pick(f, p, orientation_wxyz, 4, f->host_a), fleet_pick(f, p, angular_velocity, 3, f->host_b), 0);
    if (rc != SHC_OK) return rc;
  }
  return SHC_OK;
}
extern "C" int shc_fleet_set_pose_input(shc_fleet *f, const double *translation_velocity, const double *rotation_velocity) {
  if (!f) return fail(SHC_ERR_INVALID_ARG, "fleet is NULL");
  for (auto &p : f->parts) {
    const int rc =
        shc_engine_set_pose_input(p.engine, fleet_pick(f, p, translation_velocity, 3, f->host_a), fleet_pick(f, p, rotation_velocity, 3, f->host_b), 0);
    if (rc != SHC_OK) return rc;
  }
  return SHC_OK;
}
// per-leg inputs arrive padded: [n][max_legs][max_k]; entries beyond a morphology's (legs, k) are ignored
static int fleet_set_leg(shc_fleet *f, const double *src, int max_k, bool per_dof, int (*setter)(shc_engine *, const double *, int)) {
  if (!f) return fail(SHC_ERR_INVALID_ARG, "fleet is NULL");
  if (!src) return SHC_OK;
  for (auto &p : f->parts) {
    const shc_params &pp = f->params[p.morph];
    const int L = pp.leg_count, K = per_dof ? max_dof(pp) : max_k;
    f->host_a.resize(p.ids.size() * size_t(L) * K);
    for (size_t k = 0; k < p.ids.size(); ++k)
      for (int l = 0; l < L; ++l)
        for (int j = 0; j < K; ++j) f->host_a[(k * L + l) * K + j] = src[(size_t(p.ids[k]) * f->max_legs + l) * max_k + j];
    const int rc = setter(p.engine, f->host_a.data(), 0);
    if (rc != SHC_OK) return rc;
  }
  return SHC_OK;
}
extern "C" int shc_fleet_set_tip_force(shc_fleet *f, const double *tip_force) { return fleet_set_leg(f, tip_force, 3, false, shc_engine_set_tip_force); }
extern "C" int shc_fleet_set_joint_effort(shc_fleet *f, const double *joint_effort) {
  return fleet_set_leg(f, joint_effort, f ? f->max_dof : 0, true, shc_engine_set_joint_effort);
}
extern "C" int shc_fleet_shape(const shc_fleet *f, int *max_legs, int *max_dof) {
  if (!f) return fail(SHC_ERR_INVALID_ARG, "fleet is NULL");
  if (max_legs) *max_legs = f->max_legs;
  if (max_dof) *max_dof = f->max_dof;
  return SHC_OK;
}

// every part advances on its own stream; nothing orders one part against another
extern "C" int shc_fleet_step(shc_fleet *f, int n_cycles) {
  if (!f) return fail(SHC_ERR_INVALID_ARG, "fleet is NULL");
  for (auto &p : f->parts) {
    const int rc = shc_engine_step(p.engine, n_cycles);
    if (rc != SHC_OK) return rc;
  }
  return SHC_OK;
}
extern "C" int shc_fleet_synchronize(shc_fleet *f) {
  if (!f) return fail(SHC_ERR_INVALID_ARG, "fleet is NULL");
  for (auto &p : f->parts) {
    const int rc = shc_engine_synchronize(p.engine);
    if (rc != SHC_OK) return rc;
  }
  return SHC_OK;
}

// q / qd: [n][max_legs][max_dof] in the caller's instance order, NaN where a morphology has no such leg / joint
extern "C" int shc_fleet_get_joint_state(shc_fleet *f, double *q, double *qd) {
  if (!f) return fail(SHC_ERR_INVALID_ARG, "fleet is NULL");
  const size_t row = size_t(f->max_legs) * f->max_dof;
  for (double *dst : {q, qd})
    if (dst) std::fill(dst, dst + size_t(f->n) * row, std::nan(""));
  for (auto &p : f->parts) {
    const shc_params &pp = f->params[p.morph];
    const int L = pp.leg_count, D = max_dof(pp);
    f->host_a.resize(p.ids.size() * size_t(L) * D);
    f->host_b.resize(f->host_a.size());
    const int rc = shc_engine_get_joint_state(p.engine, q ? f->host_a.data() : nullptr, qd ? f->host_b.data() : nullptr, 0);
    if (rc != SHC_OK) return rc;
    for (size_t k = 0; k < p.ids.size(); ++k)
      for (int l = 0; l < L; ++l)
        for (int j = 0; j < D; ++j) {
          const size_t o = size_t(p.ids[k]) * row + size_t(l) * f->max_dof + j;
          if (q) q[o] = f->host_a[(k * L + l) * D + j];
          if (qd) qd[o] = f->host_b[(k * L + l) * D + j];
        }
  }
  return SHC_OK;
}
extern "C" int shc_fleet_get_walk_state(shc_fleet *f, int32_t *walk_state) {
  if (!f || !walk_state) return fail(SHC_ERR_INVALID_ARG, "NULL argument");
  for (auto &p : f->parts) {
    f->host_i.resize(p.ids.size());
    const int rc = shc_engine_get_body_state(p.engine, nullptr, nullptr, f->host_i.data(), 0);
    if (rc != SHC_OK) return rc;
    for (size_t k = 0; k < p.ids.size(); ++k) walk_state[p.ids[k]] = f->host_i[k];
  }
  return SHC_OK;
}

// Scatter of one part's joints ([rows][L][D], contiguous) into a gather buffer ([n][max_legs][max_dof]) at the caller's ids.
__global__ void fleet_place_joints_kernel(double *gather, const double *part, const int64_t *ids, int64_t rows, int L, int D, int max_legs, int max_dof) {
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= rows * L * D) return;
  const int64_t k = t / (L * D);
  const int r = int(t - k * L * D), l = r / D, j = r - l * D;
  gather[(ids[k] * max_legs + l) * max_dof + j] = part[t];
}

// The exchange step: every device ends up with the desired joint positions of ALL instances ([n][max_legs][max_dof], NaN
// padded, caller's order) in its own HBM; device_buffers[d] receives device d's buffer (owned by the fleet).  Each part
// gathers its joints on its own device and stream, places them in the local buffer, and every other device pulls the part's
// rows with one peer copy on its own gather stream (ordered behind the part by an event) and places them on its side.
// Direct peer copies rather than a ring collective: xGMI is point to point (7 links per GPU), so the (device, device) pairs use
// their own links concurrently - 1/8 of the data per link instead of the 7/8 a ring all-gather pushes through each - and a
// single-process host needs no communicator; the one-process-per-GPU host (bench.py, parallel.py) exchanges the same buffer
// with RCCL's all-gather.  Every buffer is allocated once (first call) and kept; nothing synchronises device-wide.
extern "C" int shc_fleet_all_gather_joints(shc_fleet *f, double **device_buffers) {
  if (!f) return fail(SHC_ERR_INVALID_ARG, "fleet is NULL");
  const size_t row = size_t(f->max_legs) * f->max_dof, total = size_t(f->n) * row;
  const int nd = int(f->devices.size());
  if (f->gather_stream.size() != size_t(nd)) f->gather_stream.assign(nd, nullptr);
  for (int d = 0; d < nd; ++d) {
    HIP_TRY(hipSetDevice(f->devices[d]));
    if (!f->gather_stream[d]) HIP_TRY(hipStreamCreateWithFlags(&f->gather_stream[d], hipStreamNonBlocking));
    if (!f->gather[d]) { // NaN padding (0xFF bytes are a NaN pattern), written once: the parts only ever overwrite their own entries
      HIP_TRY(hipMalloc(&f->gather[d], total * 8));
      HIP_TRY(hipMemsetAsync(f->gather[d], 0xFF, total * 8, f->gather_stream[d]));
      HIP_TRY(hipStreamSynchronize(f->gather_stream[d]));
    }
  }
  for (auto &p : f->parts) { // first call: the part's own buffers, the landing buffers on the other devices, the ids everywhere
    if (p.g_joints) continue;
    const shc_params &pp = f->params[p.morph];
    const size_t rows = p.ids.size(), elems = rows * pp.leg_count * max_dof(pp);
    HIP_TRY(hipSetDevice(p.device));
    HIP_TRY(hipMalloc(&p.g_joints, elems * 8));
    HIP_TRY(hipMalloc(&p.g_ids, rows * 8));
    HIP_TRY(hipEventCreateWithFlags(&p.g_ready, hipEventDisableTiming));
    HIP_TRY(hipMemcpy(p.g_ids, p.ids.data(), rows * 8, hipMemcpyHostToDevice));
    p.r_joints.assign(nd, nullptr);
    p.r_ids.assign(nd, nullptr);
    for (int d = 0; d < nd; ++d) {
      if (f->devices[d] == p.device) continue;
      HIP_TRY(hipSetDevice(f->devices[d]));
      HIP_TRY(hipMalloc(&p.r_joints[d], elems * 8));
      HIP_TRY(hipMalloc(&p.r_ids[d], rows * 8));
      HIP_TRY(hipMemcpy(p.r_ids[d], p.ids.data(), rows * 8, hipMemcpyHostToDevice));
    }
  }
  for (auto &p : f->parts) { // every part: joints -> its buffer -> the local gather buffer(s), on its own stream
    const shc_params &pp = f->params[p.morph];
    const int L = pp.leg_count, D = max_dof(pp);
    const int64_t rows = int64_t(p.ids.size()), threads = rows * L * D;
    HIP_TRY(hipSetDevice(p.device));
    const int rc = shc_engine_get_joint_state(p.engine, p.g_joints, nullptr, 1);
    if (rc != SHC_OK) return rc;
    HIP_TRY(hipEventRecord(p.g_ready, p.stream));
    for (int d = 0; d < nd; ++d)
      if (f->devices[d] == p.device) { // (also covers several device slots naming one device)
        fleet_place_joints_kernel<<<dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, p.stream>>>(f->gather[d], p.g_joints, p.g_ids, rows, L, D, f->max_legs,
                                                                                                   f->max_dof);
        HIP_TRY(hipGetLastError());
      }
  }
  for (auto &p : f->parts) { // every other device pulls the part's rows and places them
    const shc_params &pp = f->params[p.morph];
    const int L = pp.leg_count, D = max_dof(pp);
    const int64_t rows = int64_t(p.ids.size()), threads = rows * L * D;
    for (int d = 0; d < nd; ++d) {
      if (f->devices[d] == p.device) continue;
      HIP_TRY(hipSetDevice(f->devices[d]));
      HIP_TRY(hipStreamWaitEvent(f->gather_stream[d], p.g_ready, 0));
      HIP_TRY(hipMemcpyPeerAsync(p.r_joints[d], f->devices[d], p.g_joints, p.device, size_t(threads) * 8, f->gather_stream[d]));
      fleet_place_joints_kernel<<<dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, f->gather_stream[d]>>>(f->gather[d], p.r_joints[d], p.r_ids[d], rows, L, D,
                                                                                                            f->max_legs, f->max_dof);
      HIP_TRY(hipGetLastError());
    }
  }
  for (auto &p : f->parts) {
    HIP_TRY(hipSetDevice(p.device));
    HIP_TRY(hipStreamSynchronize(p.stream));
  }
  for (int d = 0; d < nd; ++d) {
    HIP_TRY(hipSetDevice(f->devices[d]));
    HIP_TRY(hipStreamSynchronize(f->gather_stream[d]));
  }
  if (device_buffers)
    for (int d = 0; d < nd; ++d) device_buffers[d] = f->gather[d];
  return SHC_OK;
}

// ================================================================================================ one process per GPU: the exchange over peer copies
// The all-gather of the final joint buffer (BASELINE.json north_star) without a collective library, for the one-process-per-GPU host (bench.py
// --gather peer, parallel.PeerAllGather): every rank allocates its gathered buffer here, exports it (hipIpcGetMemHandle), opens its peers'
// and writes its own shard into EVERY peer's buffer at its own offset - N - 1 copies on N - 1 streams, i.e. over the N - 1 xGMI links of the GPU
// at once (41.9 MB per link, ~0.27 ms at 2^20 octopods, against ~1.9 ms for a ring that is bound by one link).  The
// copies are ordered after the caller's stream and the caller's stream after them; a barrier of the caller's (every rank has written) closes
// the exchange.  Same boundary as shc_fleet_all_gather_joints, which does this inside one process with hipMemcpyPeerAsync.
struct PeerPool {
  std::vector<hipStream_t> streams;
  std::vector<hipEvent_t> done;
  hipEvent_t start = nullptr;
};
static PeerPool g_peer_pool[64];
static std::mutex g_peer_mutex;

extern "C" int shc_peer_alloc(int device, int64_t bytes, void **device_ptr, unsigned char *handle64) {
  if (!device_ptr || !handle64 || bytes < 1) return fail(SHC_ERR_INVALID_ARG, "shc_peer_alloc: device_ptr, handle (64 bytes) and a size");
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "the exported handle is 64 bytes");
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(hipMalloc(device_ptr, size_t(bytes)));
  hipIpcMemHandle_t h;
  const hipError_t err = hipIpcGetMemHandle(&h, *device_ptr);
  if (err != hipSuccess) {
    (void)hipFree(*device_ptr);
    *device_ptr = nullptr;
    return fail(SHC_ERR_HIP, std::string("hipIpcGetMemHandle: ") + hipGetErrorString(err) + " (HSA_ENABLE_IPC_MODE_LEGACY=0 must be set before the first HIP call)");
  }
  memcpy(handle64, &h, 64);
  return SHC_OK;
}
extern "C" int shc_peer_open(int device, const unsigned char *handle64, void **device_ptr) {
  if (!device_ptr || !handle64) return fail(SHC_ERR_INVALID_ARG, "shc_peer_open: handle and device_ptr");
  HIP_TRY(hipSetDevice(device));
  hipIpcMemHandle_t h;
  memcpy(&h, handle64, 64);
  HIP_TRY(hipIpcOpenMemHandle(device_ptr, h, hipIpcMemLazyEnablePeerAccess));
  return SHC_OK;
}
// the copy streams / events shc_peer_scatter keeps per device: released when the rank frees its own gathered buffer (the exchange's lifetime)
static void peer_pool_release(int device) {
  if (device < 0 || device >= 64) return;
  std::lock_guard<std::mutex> lock(g_peer_mutex);
  PeerPool &pool = g_peer_pool[device];
  for (size_t k = 0; k < pool.streams.size(); ++k) {
    (void)hipStreamSynchronize(pool.streams[k]);
    (void)hipEventDestroy(pool.done[k]);
    (void)hipStreamDestroy(pool.streams[k]);
  }
  pool.streams.clear();
  pool.done.clear();
  if (pool.start) (void)hipEventDestroy(pool.start);
  pool.start = nullptr;
}
extern "C" int shc_peer_close(int device, void *device_ptr, int opened) {
  if (!device_ptr) return SHC_OK;
  HIP_TRY(hipSetDevice(device));
  if (opened) {
    HIP_TRY(hipIpcCloseMemHandle(device_ptr));
  } else {
    peer_pool_release(device);
    HIP_TRY(hipFree(device_ptr));
  }
  return SHC_OK;
}
// src (this device) -> dst[k] for k < n_dst, each copy on a stream of its own; ordered after `stream`, and `stream` is ordered after all of them
extern "C" int shc_peer_scatter(int device, const void *src, int64_t bytes, void *const *dst, int n_dst, void *stream) {
  if (!src || !dst || n_dst < 1 || bytes < 1) return fail(SHC_ERR_INVALID_ARG, "shc_peer_scatter: src, destinations and a size");
  if (device < 0 || device >= 64) return fail(SHC_ERR_INVALID_ARG, "device index");
  HIP_TRY(hipSetDevice(device));
  std::lock_guard<std::mutex> lock(g_peer_mutex);
  PeerPool &pool = g_peer_pool[device];
  if (!pool.start) HIP_TRY(hipEventCreateWithFlags(&pool.start, hipEventDisableTiming));
  while (int(pool.streams.size()) < n_dst) {
    hipStream_t s;
    hipEvent_t ev;
    HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    if (const hipError_t err = hipEventCreateWithFlags(&ev, hipEventDisableTiming); err != hipSuccess) {
      (void)hipStreamDestroy(s);
      return fail(SHC_ERR_HIP, std::string("shc_peer_scatter: hipEventCreateWithFlags: ") + hipGetErrorString(err));
    }
    pool.streams.push_back(s);
    pool.done.push_back(ev);
  }
  hipStream_t main = (hipStream_t)stream;
  HIP_TRY(hipEventRecord(pool.start, main));
  for (int k = 0; k < n_dst; ++k) {
    HIP_TRY(hipStreamWaitEvent(pool.streams[k], pool.start, 0));
    HIP_TRY(hipMemcpyAsync(dst[k], src, size_t(bytes), hipMemcpyDeviceToDevice, pool.streams[k]));
    HIP_TRY(hipEventRecord(pool.done[k], pool.streams[k]));
    HIP_TRY(hipStreamWaitEvent(main, pool.done[k], 0));
  }
  return SHC_OK;
}
