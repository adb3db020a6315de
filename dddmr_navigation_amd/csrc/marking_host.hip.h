// marking_host.hip.h -- host side of the global-mode marking / clearing layer (included by
// rollout_engine.hip after the context and its helpers are defined): device state, the per-update
// launch sequence and the dddmr_rollout_marking_* entry points of include/dddmr_rollout.h.
#pragma once

#include <unordered_map>

#include "marking.hip.h"
#include "marking_fused.hip.h"

namespace {

using namespace dddmr;

constexpr uint32_t kMarkMaxObs = 1u << 20;   // points of one observation (20-bit point index inside the sort keys)

struct GridBuf {              // a PointGrid with its storage
  PointGrid g;
  uint32_t cap_cells = 0, cap_points = 0;
  uint32_t max_row = 0;       // static grids: points of the fullest (y, z) row of cells
};

struct MarkingState {
  dddmr_marking_config cfg{};
  uint32_t n_ground = 0, n_map = 0, table = 0, pool_cap = 0, max_obs = 0;
  float4 *ground_pts = nullptr, *map_pts = nullptr;
  GridBuf ground, map, obs[2];
  int prev = -1;              // obs[] entry holding pcl_msg_gbl_ of the last selfMark, -1 = none yet
  uint32_t n_prev = 0;
  float4* obs_copy[2] = {nullptr, nullptr};
  MarkStore store{};
  float4* pool_alt = nullptr;
  // second set of the per-slot arrays for the store's garbage collection (k_mk_rehash)
  unsigned long long* keys_alt = nullptr;
  uint32_t *alive_alt = nullptr, *pts_ofs_alt = nullptr, *pts_n_alt = nullptr;
  uint32_t pool_used_host = 0, n_alive_host = 0, keys_used_host = 0;
  // scratch of one update (sized for max_obs)
  uint2* gslot = nullptr;
  uint32_t* parent = nullptr;
  unsigned long long *keys_a = nullptr, *keys_b = nullptr, *keys1 = nullptr;
  uint32_t *vals_a = nullptr, *vals_b = nullptr, *flags = nullptr, *incl = nullptr, *cid_incl = nullptr;
  float4 *ds = nullptr, *proj = nullptr, *gen = nullptr;
  uint32_t *ds_first = nullptr, *pool_ofs = nullptr, *compact_sizes = nullptr, *compact_ofs = nullptr;
  ClusterArrays cl{};
  MarkCounters* counters = nullptr;        // device
  uint32_t* n_groups = nullptr;            // [2] device: groups of the 0.2 m and of the 0.1 m VoxelGrid
  void* temp = nullptr;
  size_t temp_bytes = 0;
  hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
  uint32_t seq = 0;
  // fused route (marking_fused.hip.h)
  float4* unmark_pts = nullptr;            // [pool_cap]
  float4* band_pts = nullptr;              // [2][kBandMax * kBandCap]: generator points of this update / of removed markings, by row of ground cells
  uint32_t* band_cnt = nullptr;            // [2][kBandMax], all zero between updates
  uint32_t* hi_rank = nullptr;             // [8][kFuseMaxObs - kFuseRegObs] grid builder scratch for observations past kFuseRegObs points
  uint32_t* clear_list = nullptr;          // [table]
  uint32_t* cell_count = nullptr;          // [kFuseMaxCells], all zero between updates
  bool alive_list_stale = false;           // the fused route keeps no alive list: the general route rebuilds it first
  uint32_t* ticket = nullptr;              // [2]
  MarkCounters* host_out = nullptr;        // host-mapped (hipHostMalloc), host_out_dev = its device address
  MarkCounters* host_out_dev = nullptr;
  bool counters_clean = false;             // device counters zeroed (pool fill kept) by the last fused update
  int route = -1;                          // DDDMR_MARKING_ROUTE: -1 by size, 0 general (rocPRIM), 1 fused only
  uint32_t updates_fused = 0, updates_general = 0, launches_last = 0;
  float last_clear_ms = 0.f, last_mark_ms = 0.f;
  hipStream_t own_stream = nullptr;        // the update's stream while a tick_begin is pending (else the context's)
  hipStream_t cur = nullptr;               // stream of the update in progress
  uint32_t updates_overlapped = 0;
  uint32_t grid_parts = 4;                 // DDDMR_MKF_GRIDPARTS: workgroups that build the observation grid (1, 2, 4, 8)
  bool unmark_with_groups = true;          // DDDMR_MKF_UNMARK=roots: removePCPtr blocks in the seeds' launch instead of the partitions'
  bool grid_in_lds = true;                 // DDDMR_MKF_GRID=global: counts by a count launch's global atomics instead
  uint32_t splat_parts = 0;                // DDDMR_MKF_PARTS: blocks that share a row segment's bands in the commit launch (0: estimated)
  uint32_t unmark_parts = 8;               // DDDMR_MKF_UNPARTS (tuning)
  uint32_t fuse_cells = kFuseMaxCells;     // DDDMR_MKF_CELLS: cells of the fused route's observation grid (tuning)
};

void free_grid(GridBuf& b) {
  if (b.g.cell_start) (void)hipFree(b.g.cell_start);
  if (b.g.sorted) (void)hipFree(b.g.sorted);
  b = GridBuf();
}

// Contested voxels.  When several accepted clusters of one scan have their centroid in the same voxel, the reference
// keeps the generator points of the one processed LAST, and it processes the clusters in the order
// std::sort(clusters.rbegin(), clusters.rend(), comparePointClusters) leaves them in (EuclideanClusterExtraction::
// extract): descending size, equal sizes in the order libstdc++'s introsort happens to produce.  k_mk_slots breaks
// equal sizes by cluster index; for the (rare) updates that have a contested voxel this replays the reference's sort on
// the host -- the same std::sort, on the same sizes in the same creation order (ascending first point index =
// ascending cluster index), with a comparator that compares sizes only -- and re-commits the voxels whose keeper
// differs.  The dGraph and the lethal set do not depend on the keeper (every cluster contributes its minimum); what does
// is which node set a later selfClear of the voxel resets.
int marking_fix_ties(dddmr_rollout_ctx* ctx, MarkingState* m, const MarkParams& k, const MarkStore& s, MarkCounters& out);

void marking_free(MarkingState* m) {
  if (!m) return;
  void* p[] = {m->ground_pts, m->map_pts, m->obs_copy[0], m->obs_copy[1], m->store.keys, m->store.alive, m->store.pts_ofs,
               m->store.pts_n, m->store.removed_seq, m->store.owner, m->store.alive_list, m->store.removed_list, m->store.fov_flag,
               m->store.pool, m->pool_alt, m->store.dgraph, m->store.lethal, m->keys_alt, m->alive_alt, m->pts_ofs_alt, m->pts_n_alt,
               m->gslot, m->parent, m->keys_a, m->keys_b, m->keys1, m->vals_a, m->vals_b, m->flags, m->incl, m->cid_incl, m->ds,
               m->proj, m->gen, m->ds_first, m->pool_ofs, m->compact_sizes, m->compact_ofs, m->cl.start, m->cl.size, m->cl.centroid,
               m->cl.state, m->cl.ds_count, m->cl.gen_first, m->cl.gen_count, m->cl.slot, m->cl.vkey, m->counters, m->n_groups,
               m->temp, m->unmark_pts, m->ticket, m->clear_list, m->cell_count, m->band_pts, m->band_cnt, m->hi_rank};
  for (void* q : p)
    if (q) (void)hipFree(q);
  if (m->host_out) (void)hipHostFree(m->host_out);
  free_grid(m->ground); free_grid(m->map); free_grid(m->obs[0]); free_grid(m->obs[1]);
  if (m->own_stream) (void)hipStreamDestroy(m->own_stream);
  if (m->e0) (void)hipEventDestroy(m->e0);
  if (m->e1) (void)hipEventDestroy(m->e1);
  if (m->e2) (void)hipEventDestroy(m->e2);
  delete m;
}

// Grid geometry over [lo, hi] with the wanted cell sizes, coarsened until it fits `cap_cells`.
void grid_shape(PointGrid& g, const float lo[3], const float hi[3], float cell_xy, float cell_z, uint32_t cap_cells) {
  for (;;) {
    g.nx = std::max(1, (int)std::ceil((hi[0] - lo[0]) / cell_xy) + 1);
    g.ny = std::max(1, (int)std::ceil((hi[1] - lo[1]) / cell_xy) + 1);
    g.nz = std::max(1, (int)std::ceil((hi[2] - lo[2]) / cell_z) + 1);
    if ((uint64_t)g.nx * g.ny * g.nz <= cap_cells) break;
    cell_xy *= 1.3f;
    cell_z *= 1.3f;
  }
  g.ox = lo[0]; g.oy = lo[1]; g.oz = lo[2];
  g.inv_xy = 1.0f / cell_xy;
  g.inv_z = 1.0f / cell_z;
}

// grid_cx / grid_cy / grid_cz of marking.hip.h on the host (same float operations)
int host_cx(const PointGrid& g, float x) { return std::min(std::max((int)std::floor((x - g.ox) * g.inv_xy), 0), g.nx - 1); }
int host_cy(const PointGrid& g, float y) { return std::min(std::max((int)std::floor((y - g.oy) * g.inv_xy), 0), g.ny - 1); }
int host_cz(const PointGrid& g, float z) { return std::min(std::max((int)std::floor((z - g.oz) * g.inv_z), 0), g.nz - 1); }

int grid_alloc(dddmr_rollout_ctx* ctx, GridBuf& b, uint32_t cap_cells, uint32_t cap_points) {
  b.cap_cells = cap_cells;
  b.cap_points = cap_points;
  HIPCHK(ctx, hipMalloc(&b.g.cell_start, ((size_t)cap_cells + 1) * sizeof(uint32_t)));
  HIPCHK(ctx, hipMalloc(&b.g.sorted, (size_t)std::max<uint32_t>(cap_points, 1) * sizeof(float4)));
  return DDDMR_OK;
}

// count -> exclusive scan -> scatter on `stream`; `counts` doubles as cell_start
int grid_build(dddmr_rollout_ctx* ctx, MarkingState* m, GridBuf& b, const float4* pts, uint32_t n, uint2* slot, hipStream_t stream) {
  PointGrid& g = b.g;
  g.n = n;
  const size_t cells = (size_t)g.nx * g.ny * g.nz;
  HIPCHK(ctx, hipMemsetAsync(g.cell_start, 0, (cells + 1) * sizeof(uint32_t), stream));
  if (n == 0) return DDDMR_OK;
  hipLaunchKernelGGL(k_grid_count, dim3((n + 255) / 256), dim3(256), 0, stream, g, pts, g.cell_start, slot);
  size_t need = m->temp_bytes;
  HIPCHK(ctx, rocprim::exclusive_scan(m->temp, need, g.cell_start, g.cell_start, 0u, cells + 1, rocprim::plus<uint32_t>(), stream));
  hipLaunchKernelGGL(k_grid_scatter, dim3((n + 255) / 256), dim3(256), 0, stream, g, pts, slot);
  return DDDMR_OK;
}

void quat_rotate_z(const double q[4], double out[3]) {   // tf2::quatRotate(q, (0, 0, 1)); q = x y z w
  // q * v
  const double ax = q[1] * 1.0, ay = -q[0] * 1.0, az = q[3] * 1.0, aw = -q[2] * 1.0;   // (w*0 + y*1 - z*0, w*0 + z*0 - x*1, w*1 + x*0 - y*0, -x*0 - y*0 - z*1)
  // (q * v) * q^-1, q^-1 = (-x, -y, -z, w)
  const double bx = -q[0], by = -q[1], bz = -q[2], bw = q[3];
  out[0] = aw * bx + ax * bw + ay * bz - az * by;
  out[1] = aw * by + ay * bw + az * bx - ax * bz;
  out[2] = aw * bz + az * bw + ax * by - ay * bx;
}

// Eigen quaternion of a rotation matrix (row-major R), as tf2::eigenToTransform forms trans_gbl2s_'s rotation
void rot_to_quat(const double R[9], double q[4]) {
  double t = R[0] + R[4] + R[8];
  if (t > 0.0) {
    t = std::sqrt(t + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (R[7] - R[5]) * t;
    q[1] = (R[2] - R[6]) * t;
    q[2] = (R[3] - R[1]) * t;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[4 * i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (R[3 * k + j] - R[3 * j + k]) * t;
    q[j] = (R[3 * j + i] + R[3 * i + j]) * t;
    q[k] = (R[3 * k + i] + R[3 * i + k]) * t;
  }
}
// tf2 Matrix3x3::setRotation(q), row-major
void tf2_set_rotation(const double q[4], double R[9]) {
  const double d = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  const double s = 2.0 / d;
  const double xs = q[0] * s, ys = q[1] * s, zs = q[2] * s;
  const double wx = q[3] * xs, wy = q[3] * ys, wz = q[3] * zs;
  const double xx = q[0] * xs, xy = q[0] * ys, xz = q[0] * zs;
  const double yy = q[1] * ys, yz = q[1] * zs, zz = q[2] * zs;
  R[0] = 1.0 - (yy + zz); R[1] = xy - wz; R[2] = xz + wy;
  R[3] = xy + wz; R[4] = 1.0 - (xx + zz); R[5] = yz - wx;
  R[6] = xz - wy; R[7] = yz + wx; R[8] = 1.0 - (xx + yy);
}

int upload_static(dddmr_rollout_ctx* ctx, MarkingState* m, GridBuf& b, float4** dev, const float* xyz, size_t n, size_t stride_bytes,
                  float cell_xy, float cell_z) {
  std::vector<float4> h(std::max<size_t>(n, 1));
  float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
  const size_t sf = stride_bytes / sizeof(float);
  for (size_t i = 0; i < n; ++i) {
    const float* p = xyz + i * sf;
    h[i] = make_float4(p[0], p[1], p[2], 0.f);
    for (int a = 0; a < 3; ++a) {
      lo[a] = i ? std::min(lo[a], p[a]) : p[a];
      hi[a] = i ? std::max(hi[a], p[a]) : p[a];
    }
  }
  HIPCHK(ctx, hipMalloc(dev, h.size() * sizeof(float4)));
  HIPCHK(ctx, hipMemcpy(*dev, h.data(), h.size() * sizeof(float4), hipMemcpyHostToDevice));
  grid_shape(b.g, lo, hi, cell_xy, cell_z, 1u << 22);
  const int rc = grid_alloc(ctx, b, (uint32_t)((size_t)b.g.nx * b.g.ny * b.g.nz), (uint32_t)n);
  if (rc != DDDMR_OK) return rc;
  {
    std::vector<uint32_t> rows((size_t)b.g.ny * b.g.nz, 0u);
    for (size_t i = 0; i < n; ++i) ++rows[(size_t)host_cz(b.g, h[i].z) * b.g.ny + host_cy(b.g, h[i].y)];
    b.max_row = 0;
    for (uint32_t r : rows) b.max_row = std::max(b.max_row, r);
  }
  uint2* slot = nullptr;
  HIPCHK(ctx, hipMalloc(&slot, std::max<size_t>(n, 1) * sizeof(uint2)));
  const int rb = grid_build(ctx, m, b, *dev, (uint32_t)n, slot, ctx->stream);
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  (void)hipFree(slot);
  return rb;
}

int marking_fix_ties(dddmr_rollout_ctx* ctx, MarkingState* m, const MarkParams& k, const MarkStore& s, MarkCounters& out) {
  const uint32_t nc = out.n_clusters;
  if (nc == 0) return DDDMR_OK;
  hipStream_t st = m->cur;
  std::vector<uint32_t> size(nc), state(nc), slot(nc);
  HIPCHK(ctx, hipMemcpyAsync(size.data(), m->cl.size, nc * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  HIPCHK(ctx, hipMemcpyAsync(state.data(), m->cl.state, nc * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  HIPCHK(ctx, hipMemcpyAsync(slot.data(), m->cl.slot, nc * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  HIPCHK(ctx, hipStreamSynchronize(st));
  // what extractEuclideanClusters hands to the sort: the clusters of at least min_cluster_size points, in creation order
  struct Item { uint32_t size, ci; };
  std::vector<Item> order;
  order.reserve(nc);
  for (uint32_t ci = 0; ci < nc; ++ci)     // (size 0: a point index that seeds no cluster, fused route)
    if (size[ci] > 0 && (int)size[ci] >= m->cfg.euclidean_cluster_extraction_min_cluster_size) order.push_back(Item{size[ci], ci});
  std::sort(order.rbegin(), order.rend(), [](const Item& a, const Item& b) { return a.size < b.size; });
  // per contested voxel: the accepted cluster the reference processes last, against the one the device kept
  struct Keep { uint32_t ref_ci, dev_ci, dev_size, claims; };
  std::unordered_map<uint32_t, Keep> keep;
  for (const Item& it : order) {                      // (processing order)
    if (state[it.ci] != 2u) continue;
    auto ins = keep.insert(std::make_pair(slot[it.ci], Keep{it.ci, it.ci, it.size, 1u}));
    if (ins.second) continue;
    Keep& kp = ins.first->second;
    kp.ref_ci = it.ci;
    ++kp.claims;
    if (it.size < kp.dev_size || (it.size == kp.dev_size && it.ci > kp.dev_ci)) { kp.dev_ci = it.ci; kp.dev_size = it.size; }   // k_mk_slots' priority
  }
  std::vector<uint2> fix;
  for (const auto& kv : keep)
    if (kv.second.claims > 1 && kv.second.ref_ci != kv.second.dev_ci) fix.push_back(make_uint2(kv.first, kv.second.ref_ci));
  if (fix.empty()) return DDDMR_OK;
  // (vals_a is scratch of the update that just finished: >= max_obs words)
  uint2* fix_dev = reinterpret_cast<uint2*>(m->vals_b);
  if (fix.size() * 2 > (size_t)m->max_obs) return fail(ctx, DDDMR_ERR_CAPACITY, "marking_update: %zu contested voxels", fix.size());
  HIPCHK(ctx, hipMemcpyAsync(fix_dev, fix.data(), fix.size() * sizeof(uint2), hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_mk_fix_owner, dim3((unsigned)((fix.size() + 3) / 4)), dim3(256), 0, st, k, (uint32_t)fix.size(), fix_dev,
                     m->gen, m->cl, s, m->counters);
  MarkCounters after{};                                // (on the fused route the device copy holds only what k_mk_fix_owner touches)
  HIPCHK(ctx, hipMemcpyAsync(&after, m->counters, sizeof(after), hipMemcpyDeviceToHost, st));
  HIPCHK(ctx, hipStreamSynchronize(st));               // (also keeps `fix` alive until the copy has run)
  HIPCHK(ctx, hipGetLastError());
  out.pool_used = after.pool_used;
  out.overflow |= after.overflow;
  if (after.overflow) m->counters_clean = false;
  m->pool_used_host = out.pool_used;
  return DDDMR_OK;
}

int marking_reset_locked(dddmr_rollout_ctx* ctx);

}  // namespace

extern "C" {

int dddmr_rollout_marking_create(dddmr_rollout_ctx* ctx, const dddmr_marking_config* cfg, const float* ground_xyz,
                                 size_t n_ground, size_t ground_stride_bytes, const float* map_xyz, size_t n_map,
                                 size_t map_stride_bytes) {
  if (!ctx || !cfg) return DDDMR_ERR_BAD_ARG;
  if ((n_ground && (!ground_xyz || ground_stride_bytes < 12 || ground_stride_bytes % 4)) ||
      (n_map && (!map_xyz || map_stride_bytes < 12 || map_stride_bytes % 4)))
    return fail(ctx, DDDMR_ERR_BAD_ARG, "marking_create: bad cloud pointer / stride");
  if (!(cfg->xy_resolution > 0) || !(cfg->height_resolution > 0) || !(cfg->euclidean_cluster_extraction_tolerance > 0) ||
      !(cfg->inflation_radius > 0) || cfg->max_markings == 0 || cfg->max_cluster_points == 0)
    return fail(ctx, DDDMR_ERR_BAD_ARG, "marking_create: resolutions, tolerance, inflation radius and capacities must be positive");
  if (n_ground >= (1u << 30) || cfg->max_markings > (1u << 24)) return fail(ctx, DDDMR_ERR_CAPACITY, "marking_create: too large");
  // the sort keys of an update carry the observation point index in 20 bits ((root << 20) | i, cluster id << 42)
  if (ctx->cfg.max_points > kMarkMaxObs)
    return fail(ctx, DDDMR_ERR_CAPACITY, "marking_create: the context's max_points %u exceeds the layer's %u-point observations",
                ctx->cfg.max_points, kMarkMaxObs);
  std::lock_guard<std::mutex> tk(ctx->tick_mu);
  if (ctx->pend.active) return fail(ctx, DDDMR_ERR_STATE, "marking_create while a tick_begin is pending");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (ctx->marking) { marking_free(ctx->marking); ctx->marking = nullptr; }
  auto* m = new MarkingState();
  ctx->marking = m;
  m->cfg = *cfg;
  m->n_ground = (uint32_t)n_ground;
  m->n_map = (uint32_t)n_map;
  m->max_obs = ctx->cfg.max_points;
  uint32_t table = 1024;
  while (table < 2 * cfg->max_markings) table <<= 1;
  m->table = table;
  m->pool_cap = cfg->max_cluster_points;
  auto init = [&]() -> int {
    const size_t N = m->max_obs;
    // rocPRIM temporary storage: the largest request among the sorts and scans used below
    {
      size_t a = 0, b = 0, c = 0, d = 0;
      unsigned long long* k = nullptr;
      uint32_t* v = nullptr;
      HIPCHK(ctx, rocprim::radix_sort_keys(nullptr, a, k, k, N, 0, 40, ctx->stream));
      HIPCHK(ctx, rocprim::radix_sort_pairs(nullptr, b, k, k, v, v, N, 0, 62, ctx->stream));
      HIPCHK(ctx, rocprim::exclusive_scan(nullptr, c, v, v, 0u, (size_t)(1u << 22) + 1, rocprim::plus<uint32_t>(), ctx->stream));
      HIPCHK(ctx, rocprim::inclusive_scan(nullptr, d, v, v, std::max<size_t>(N, m->table), rocprim::plus<uint32_t>(), ctx->stream));
      m->temp_bytes = std::max({a, b, c, d, (size_t)4096}) + 256;
      HIPCHK(ctx, hipMalloc(&m->temp, m->temp_bytes));
    }
    int rc = upload_static(ctx, m, m->ground, &m->ground_pts, ground_xyz, n_ground, ground_stride_bytes, 0.5f, 1e6f);
    if (rc != DDDMR_OK) return rc;
    rc = upload_static(ctx, m, m->map, &m->map_pts, map_xyz, n_map, map_stride_bytes, 0.25f, 0.25f);
    if (rc != DDDMR_OK) return rc;
    for (int i = 0; i < 2; ++i) {
      rc = grid_alloc(ctx, m->obs[i], 1u << 21, (uint32_t)N);
      if (rc != DDDMR_OK) return rc;
      HIPCHK(ctx, hipMalloc(&m->obs_copy[i], N * sizeof(float4)));
    }
    MarkStore& s = m->store;
    HIPCHK(ctx, hipMalloc(&s.keys, (size_t)table * sizeof(unsigned long long)));
    HIPCHK(ctx, hipMalloc(&s.alive, (size_t)table * sizeof(uint32_t)));
    HIPCHK(ctx, hipMalloc(&s.pts_ofs, (size_t)table * sizeof(uint32_t)));
    HIPCHK(ctx, hipMalloc(&s.pts_n, (size_t)table * sizeof(uint32_t)));
    HIPCHK(ctx, hipMalloc(&s.removed_seq, (size_t)table * sizeof(uint32_t)));
    HIPCHK(ctx, hipMalloc(&s.owner, (size_t)table * sizeof(unsigned long long)));
    HIPCHK(ctx, hipMalloc(&s.alive_list, (size_t)table * sizeof(uint32_t)));
    HIPCHK(ctx, hipMalloc(&s.removed_list, (size_t)table * sizeof(uint32_t)));
    HIPCHK(ctx, hipMalloc(&s.fov_flag, (size_t)table * sizeof(uint32_t)));
    HIPCHK(ctx, hipMalloc(&s.pool, (size_t)m->pool_cap * sizeof(float4)));
    HIPCHK(ctx, hipMalloc(&m->pool_alt, (size_t)m->pool_cap * sizeof(float4)));
    HIPCHK(ctx, hipMalloc(&m->keys_alt, (size_t)table * sizeof(unsigned long long)));
    HIPCHK(ctx, hipMalloc(&m->alive_alt, (size_t)table * sizeof(uint32_t)));
    HIPCHK(ctx, hipMalloc(&m->pts_ofs_alt, (size_t)table * sizeof(uint32_t)));
    HIPCHK(ctx, hipMalloc(&m->pts_n_alt, (size_t)table * sizeof(uint32_t)));
    HIPCHK(ctx, hipMalloc(&s.dgraph, ((size_t)n_ground + 1) * sizeof(double)));
    HIPCHK(ctx, hipMalloc(&s.lethal, (size_t)n_ground + 1));
    HIPCHK(ctx, hipMalloc(&m->gslot, N * sizeof(uint2)));
    HIPCHK(ctx, hipMalloc(&m->parent, N * sizeof(uint32_t)));
    HIPCHK(ctx, hipMalloc(&m->keys_a, N * sizeof(unsigned long long)));
    HIPCHK(ctx, hipMalloc(&m->keys_b, N * sizeof(unsigned long long)));
    HIPCHK(ctx, hipMalloc(&m->keys1, N * sizeof(unsigned long long)));
    HIPCHK(ctx, hipMalloc(&m->vals_a, N * sizeof(uint32_t)));
    HIPCHK(ctx, hipMalloc(&m->vals_b, N * sizeof(uint32_t)));
    HIPCHK(ctx, hipMalloc(&m->flags, N * sizeof(uint32_t)));
    HIPCHK(ctx, hipMalloc(&m->incl, N * sizeof(uint32_t)));
    HIPCHK(ctx, hipMalloc(&m->cid_incl, N * sizeof(uint32_t)));
    HIPCHK(ctx, hipMalloc(&m->ds, N * sizeof(float4)));
    HIPCHK(ctx, hipMalloc(&m->proj, N * sizeof(float4)));
    HIPCHK(ctx, hipMalloc(&m->gen, N * sizeof(float4)));
    HIPCHK(ctx, hipMalloc(&m->ds_first, N * sizeof(uint32_t)));
    HIPCHK(ctx, hipMalloc(&m->pool_ofs, N * sizeof(uint32_t)));
    HIPCHK(ctx, hipMalloc(&m->compact_sizes, (size_t)table * sizeof(uint32_t)));
    HIPCHK(ctx, hipMalloc(&m->compact_ofs, (size_t)table * sizeof(uint32_t)));
    HIPCHK(ctx, hipMalloc(&m->cl.start, (N + 1) * sizeof(uint32_t)));
    HIPCHK(ctx, hipMalloc(&m->cl.size, N * sizeof(uint32_t)));
    HIPCHK(ctx, hipMalloc(&m->cl.centroid, N * sizeof(float4)));
    HIPCHK(ctx, hipMalloc(&m->cl.state, N * sizeof(uint32_t)));
    HIPCHK(ctx, hipMalloc(&m->cl.ds_count, N * sizeof(uint32_t)));
    HIPCHK(ctx, hipMalloc(&m->cl.gen_first, N * sizeof(uint32_t)));
    HIPCHK(ctx, hipMalloc(&m->cl.gen_count, N * sizeof(uint32_t)));
    HIPCHK(ctx, hipMalloc(&m->cl.slot, N * sizeof(uint32_t)));
    HIPCHK(ctx, hipMalloc(&m->cl.vkey, 3 * N * sizeof(int)));
    HIPCHK(ctx, hipMalloc(&m->counters, sizeof(MarkCounters)));
    HIPCHK(ctx, hipMalloc(&m->n_groups, 2 * sizeof(uint32_t)));
    HIPCHK(ctx, hipMalloc(&m->unmark_pts, (size_t)m->pool_cap * sizeof(float4)));
    HIPCHK(ctx, hipMalloc(&m->band_pts, (size_t)2 * kBandMax * kBandCap * sizeof(float4)));
    HIPCHK(ctx, hipMalloc(&m->band_cnt, (size_t)2 * kBandMax * sizeof(uint32_t)));
    HIPCHK(ctx, hipMalloc(&m->hi_rank, (size_t)8 * (kFuseMaxObs - kFuseRegObs) * sizeof(uint32_t)));
    HIPCHK(ctx, hipMemset(m->band_cnt, 0, (size_t)2 * kBandMax * sizeof(uint32_t)));
    HIPCHK(ctx, hipMalloc(&m->clear_list, (size_t)table * sizeof(uint32_t)));
    HIPCHK(ctx, hipMalloc(&m->cell_count, (size_t)kFuseMaxCells * sizeof(uint32_t)));
    HIPCHK(ctx, hipMemset(m->cell_count, 0, (size_t)kFuseMaxCells * sizeof(uint32_t)));
    HIPCHK(ctx, hipMalloc(&m->ticket, 34 * sizeof(uint32_t)));
    HIPCHK(ctx, hipMemset(m->ticket, 0, 34 * sizeof(uint32_t)));
    HIPCHK(ctx, hipMemset(m->counters, 0, sizeof(MarkCounters)));
    HIPCHK(ctx, hipHostMalloc(reinterpret_cast<void**>(&m->host_out), sizeof(MarkCounters), hipHostMallocMapped));
    HIPCHK(ctx, hipHostGetDevicePointer(reinterpret_cast<void**>(&m->host_out_dev), m->host_out, 0));
    if (const char* e = std::getenv("DDDMR_MKF_UNMARK")) m->unmark_with_groups = std::strcmp(e, "roots") != 0;
    if (const char* e = std::getenv("DDDMR_MKF_GRIDPARTS")) { const int v = std::atoi(e); if (v == 1 || v == 2 || v == 4 || v == 8) m->grid_parts = (uint32_t)v; }
    if (const char* e = std::getenv("DDDMR_MKF_GRID")) m->grid_in_lds = std::strcmp(e, "global") != 0;
    if (const char* e = std::getenv("DDDMR_MKF_PARTS")) m->splat_parts = (uint32_t)std::min(128, std::max(0, std::atoi(e)));
    if (const char* e = std::getenv("DDDMR_MKF_UNPARTS")) m->unmark_parts = (uint32_t)std::min(64, std::max(1, std::atoi(e)));
    if (const char* e = std::getenv("DDDMR_MKF_CELLS")) m->fuse_cells = std::min<uint32_t>(kFuseMaxCells, std::max(4096, std::atoi(e)));
    if (const char* e = std::getenv("DDDMR_MARKING_ROUTE")) m->route = std::strcmp(e, "general") == 0 ? 0 : (std::strcmp(e, "fused") == 0 ? 1 : -1);
    HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_mkf_groups), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPartLdsBytes));
    HIPCHK(ctx, hipEventCreate(&m->e0));
    HIPCHK(ctx, hipEventCreate(&m->e1));
    HIPCHK(ctx, hipEventCreate(&m->e2));
    return DDDMR_OK;
  };
  int rc = init();
  if (rc == DDDMR_OK) rc = marking_reset_locked(ctx);
  if (rc != DDDMR_OK) { marking_free(m); ctx->marking = nullptr; }
  return rc;
}

int dddmr_rollout_marking_reset(dddmr_rollout_ctx* ctx) {
  if (!ctx) return DDDMR_ERR_BAD_ARG;
  std::lock_guard<std::mutex> tk(ctx->tick_mu);
  return marking_reset_locked(ctx);
}

}  // extern "C"

namespace {
// MultiLayerSpinningLidar::resetdGraph (:831-839): empty store, dGraph = max_obstacle_distance on keys 0..n_ground
int marking_reset_locked(dddmr_rollout_ctx* ctx) {
  MarkingState* m = ctx->marking;
  if (!m) return fail(ctx, DDDMR_ERR_STATE, "marking_reset before marking_create");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  MarkStore& s = m->store;
  const size_t t = m->table;
  HIPCHK(ctx, hipMemsetAsync(s.keys, 0, t * sizeof(unsigned long long), ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(s.alive, 0, t * sizeof(uint32_t), ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(s.pts_ofs, 0, t * sizeof(uint32_t), ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(s.pts_n, 0, t * sizeof(uint32_t), ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(s.removed_seq, 0, t * sizeof(uint32_t), ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(s.owner, 0, t * sizeof(unsigned long long), ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(s.lethal, 0, (size_t)m->n_ground + 1, ctx->stream));
  hipLaunchKernelGGL(k_mk_fill_dgraph, dim3((m->n_ground + 1 + 255) / 256), dim3(256), 0, ctx->stream, m->n_ground + 1, s.dgraph,
                     m->cfg.max_obstacle_distance);
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  m->pool_used_host = 0;
  m->n_alive_host = 0;
  m->keys_used_host = 0;
  m->counters_clean = false;
  m->alive_list_stale = false;               // (nothing alive: the empty list is right)
  // (pcl_msg_gbl_ is untouched by resetdGraph: the previous observation stays)
  return DDDMR_OK;
}

struct UpdateFrame {          // host-side inputs of one update besides the kernel parameters
  MarkParams k;
  double Rb[9];               // rotation of T_gbl_base
  double t_gb[3];
};

#define MK_LAUNCH(m, ...) do { hipLaunchKernelGGL(__VA_ARGS__); ++(m)->launches_last; } while (0)

// Store garbage collection when half the table holds keys and a good part of them is dead; pool compaction when half
// of the pool is garbage-or-used.  Both rare; both leave the device counters consistent for either route.
int store_maintenance(dddmr_rollout_ctx* ctx, MarkingState* m, hipStream_t st) {
  MarkStore& s = m->store;
  if (m->keys_used_host > m->table / 2 && m->keys_used_host > m->n_alive_host + m->table / 8) {
    const size_t t = m->table;
    HIPCHK(ctx, hipMemsetAsync(m->keys_alt, 0, t * sizeof(unsigned long long), st));
    HIPCHK(ctx, hipMemsetAsync(m->alive_alt, 0, t * sizeof(uint32_t), st));
    HIPCHK(ctx, hipMemsetAsync(m->pts_ofs_alt, 0, t * sizeof(uint32_t), st));
    HIPCHK(ctx, hipMemsetAsync(m->pts_n_alt, 0, t * sizeof(uint32_t), st));
    MK_LAUNCH(m, k_mk_rehash, dim3((m->table + 255) / 256), dim3(256), 0, st, m->table - 1, s, m->keys_alt, m->alive_alt,
              m->pts_ofs_alt, m->pts_n_alt, m->counters);
    std::swap(s.keys, m->keys_alt);
    std::swap(s.alive, m->alive_alt);
    std::swap(s.pts_ofs, m->pts_ofs_alt);
    std::swap(s.pts_n, m->pts_n_alt);
    m->keys_used_host = m->n_alive_host;
  }
  if (m->pool_used_host > m->pool_cap / 2) {
    MK_LAUNCH(m, k_mk_compact_sizes, dim3((m->table + 255) / 256), dim3(256), 0, st, m->table, s, m->compact_sizes);
    size_t tb = m->temp_bytes;
    HIPCHK(ctx, rocprim::exclusive_scan(m->temp, tb, m->compact_sizes, m->compact_ofs, 0u, (size_t)m->table, rocprim::plus<uint32_t>(), st));
    HIPCHK(ctx, hipMemsetAsync(&m->counters->pool_used, 0, sizeof(uint32_t), st));
    MK_LAUNCH(m, k_mk_compact_move, dim3((m->table + 3) / 4), dim3(256), 0, st, m->table, s, m->compact_ofs, m->pool_alt, m->counters);
    std::swap(s.pool, m->pool_alt);
  }
  return DDDMR_OK;
}

// the crop box of the feed (base frame |x|,|y| <= window, z in [0, marking_height]) in the global frame, + 0.3 m
void crop_box(const MarkingState* m, const UpdateFrame& f, float lo[3], float hi[3]) {
  const dddmr_marking_config& c = m->cfg;
  for (int a = 0; a < 3; ++a) { lo[a] = 3.4e38f; hi[a] = -3.4e38f; }
  for (int corner = 0; corner < 8; ++corner) {
    const double bx = (corner & 1) ? c.perception_window_size : -c.perception_window_size;
    const double by = (corner & 2) ? c.perception_window_size : -c.perception_window_size;
    const double bz = (corner & 4) ? c.marking_height : 0.0;
    for (int a = 0; a < 3; ++a) {
      const float v = (float)(f.Rb[3 * a] * bx + f.Rb[3 * a + 1] * by + f.Rb[3 * a + 2] * bz + f.t_gb[a]);
      lo[a] = std::min(lo[a], v - 0.3f);
      hi[a] = std::max(hi[a], v + 0.3f);
    }
  }
}
// grid over it
void obs_grid_shape(const MarkingState* m, const UpdateFrame& f, uint32_t cap_cells, PointGrid& g, float lo[3], float hi[3]) {
  crop_box(m, f, lo, hi);
  const float cell = std::max(0.1f, f.k.tol);
  grid_shape(g, lo, hi, cell, cell, cap_cells);
}

// selfMark of this observation, general route: rocPRIM sorts, any observation size (marking.hip.h)
int mark_general(dddmr_rollout_ctx* ctx, MarkingState* m, const UpdateFrame& f, const float4* obs, uint32_t n_obs, hipStream_t st) {
  const MarkParams& k = f.k;
  MarkStore& s = m->store;
  const int cur = m->prev >= 0 ? 1 - m->prev : 0;
  GridBuf& gb = m->obs[cur];
  float lo[3], hi[3];
  obs_grid_shape(m, f, gb.cap_cells, gb.g, lo, hi);
  const float4* pts = obs;                   // (the cloud buffer stays pinned until the update returns)
  int rc = grid_build(ctx, m, gb, pts, n_obs, m->gslot, st);
  m->launches_last += 5;                     // memset, count, rocPRIM scan (2 kernels), scatter
  if (rc != DDDMR_OK) return rc;
  const dim3 pb((n_obs + 255) / 256), cb((n_obs + 63) / 64);
  // Euclidean clusters
  MK_LAUNCH(m, k_mk_cc_init, pb, dim3(256), 0, st, n_obs, m->parent);
  MK_LAUNCH(m, k_mk_cc_union, pb, dim3(256), 0, st, k, gb.g, pts, m->parent);
  MK_LAUNCH(m, k_mk_cc_keys, pb, dim3(256), 0, st, n_obs, m->parent, m->keys_a);
  size_t tb = m->temp_bytes;
  HIPCHK(ctx, rocprim::radix_sort_keys(m->temp, tb, m->keys_a, m->keys1, (size_t)n_obs, 0, 40, st));
  MK_LAUNCH(m, k_mk_flags, pb, dim3(256), 0, st, n_obs, m->keys1, 20, m->flags);
  tb = m->temp_bytes;
  HIPCHK(ctx, rocprim::inclusive_scan(m->temp, tb, m->flags, m->cid_incl, (size_t)n_obs, rocprim::plus<uint32_t>(), st));
  MK_LAUNCH(m, k_mk_cluster_starts, pb, dim3(256), 0, st, n_obs, m->flags, m->cid_incl, m->cl, m->counters);
  MK_LAUNCH(m, k_mk_cluster_stage1, cb, dim3(64), 0, st, k, m->counters, m->cl, m->keys1, pts, m->ground.g);
  // 0.2 m VoxelGrid of every surviving cluster: stable sort by (cluster, voxel), one lane per voxel
  // (voxel indices are keyed relative to the window's centre - half the key range: an observation handed over uncropped may reach far
  // beyond the window -- a margin of 16 voxels around the crop box used to drop such points without a word)
  const float mid[3] = {0.5f * (lo[0] + hi[0]), 0.5f * (lo[1] + hi[1]), 0.5f * (lo[2] + hi[2])};
  const int ox2 = (int)std::floor(mid[0] / 0.2f) - kVgHalfXY, oy2 = (int)std::floor(mid[1] / 0.2f) - kVgHalfXY, oz2 = (int)std::floor(mid[2] / 0.2f) - kVgHalfZ;
  MK_LAUNCH(m, k_mk_ds_keys, pb, dim3(256), 0, st, k, m->keys1, m->cid_incl, m->cl, pts, ox2, oy2, oz2, m->keys_a, m->vals_a, m->counters);
  tb = m->temp_bytes;
  HIPCHK(ctx, rocprim::radix_sort_pairs(m->temp, tb, m->keys_a, m->keys_b, m->vals_a, m->vals_b, (size_t)n_obs, 0, 62, st));
  MK_LAUNCH(m, k_mk_flags, pb, dim3(256), 0, st, n_obs, m->keys_b, 0, m->flags);
  tb = m->temp_bytes;
  HIPCHK(ctx, rocprim::inclusive_scan(m->temp, tb, m->flags, m->incl, (size_t)n_obs, rocprim::plus<uint32_t>(), st));
  HIPCHK(ctx, hipMemsetAsync(m->ds_first, 0xFF, (size_t)n_obs * sizeof(uint32_t), st));
  MK_LAUNCH(m, k_mk_group_reduce, cb, dim3(64), 0, st, n_obs, m->keys_b, m->vals_b, m->flags, m->incl, 0, m->keys1, pts, m->ds,
            m->cl.ds_count, m->ds_first, m->n_groups);
  MK_LAUNCH(m, k_mk_cluster_stage2, cb, dim3(64), 0, st, k, m->counters, m->cl, m->map.g, m->n_map);
  // projection on the base plane + 0.1 m VoxelGrid of the accepted clusters -> generator points
  const int ox3 = (int)std::floor(mid[0] / 0.1f) - kVgHalfXY, oy3 = (int)std::floor(mid[1] / 0.1f) - kVgHalfXY, oz3 = (int)std::floor(mid[2] / 0.1f) - kVgHalfZ;
  MK_LAUNCH(m, k_mk_proj_keys, pb, dim3(256), 0, st, k, m->n_groups, m->ds, m->cl, ox3, oy3, oz3, m->proj, m->keys_a, m->vals_a, n_obs, m->counters);
  tb = m->temp_bytes;
  HIPCHK(ctx, rocprim::radix_sort_pairs(m->temp, tb, m->keys_a, m->keys_b, m->vals_a, m->vals_b, (size_t)n_obs, 0, 62, st));
  MK_LAUNCH(m, k_mk_flags, pb, dim3(256), 0, st, n_obs, m->keys_b, 0, m->flags);
  tb = m->temp_bytes;
  HIPCHK(ctx, rocprim::inclusive_scan(m->temp, tb, m->flags, m->incl, (size_t)n_obs, rocprim::plus<uint32_t>(), st));
  MK_LAUNCH(m, k_mk_group_reduce, cb, dim3(64), 0, st, n_obs, m->keys_b, m->vals_b, m->flags, m->incl, 1, m->keys1, m->proj, m->gen,
            m->cl.gen_count, m->cl.gen_first, m->n_groups + 1);
  m->launches_last += 3 * 10 + 2 * 3 + 1;   // rocPRIM: three sorts (block sort + ~8 merge passes + id wrapper), three scans, memset
  // addPCPtr
  MK_LAUNCH(m, k_mk_slots, cb, dim3(64), 0, st, k, m->counters, m->cl, s, m->counters);
  MK_LAUNCH(m, k_mk_commit, cb, dim3(64), 0, st, k, m->counters, m->cl, s, m->counters, m->pool_ofs);
  MK_LAUNCH(m, k_mk_dgraph, dim3((n_obs + 3) / 4), dim3(256), 0, st, k, m->n_groups + 1, m->gen, m->cl, m->pool_ofs, s, m->ground.g);
  m->prev = cur;                                                               // pcl_msg_gbl_ of this selfMark
  m->n_prev = n_obs;
  return DDDMR_OK;
}

// one update on the general route (~55 launches for a 6 k-point observation)
int update_general(dddmr_rollout_ctx* ctx, MarkingState* m, const UpdateFrame& f, const float4* obs, uint32_t n_obs, bool timed,
                   MarkCounters& out) {
  hipStream_t st = m->cur;
  const MarkParams& k = f.k;
  MarkStore& s = m->store;
  MarkCounters zero{};
  zero.pool_used = m->pool_used_host;
  HIPCHK(ctx, hipMemcpyAsync(m->counters, &zero, sizeof(zero), hipMemcpyHostToDevice, st));   // (pageable source: copied before return)
  m->counters_clean = false;
  if (timed) HIPCHK(ctx, hipEventRecord(m->e0, st));
  int rc = store_maintenance(ctx, m, st);
  if (rc != DDDMR_OK) return rc;
  if (m->alive_list_stale) {                 // the last update ran fused: what selfClear walks has to be listed first
    MK_LAUNCH(m, k_mk_finish, dim3((m->table + 255) / 256), dim3(256), 0, st, k, s, m->counters);
    HIPCHK(ctx, hipMemsetAsync(&m->counters->n_alive, 0, sizeof(uint32_t), st));
    m->alive_list_stale = false;
  }
  // ---- selfClear against the previous observation ----
  const PointGrid empty_grid = m->obs[0].g;
  const PointGrid& prev_grid = m->prev >= 0 ? m->obs[m->prev].g : empty_grid;
  if (m->n_alive_host > 0) {
    MK_LAUNCH(m, k_mk_fov, dim3((m->n_alive_host + 255) / 256), dim3(256), 0, st, k, s, m->counters);
    MK_LAUNCH(m, k_mk_clear, dim3((m->n_alive_host + 3) / 4), dim3(256), 0, st, k, s, prev_grid, m->counters);
    MK_LAUNCH(m, k_mk_unmark, dim3((m->n_alive_host + 3) / 4), dim3(256), 0, st, k, s, m->ground.g, m->counters);
  }
  if (timed) HIPCHK(ctx, hipEventRecord(m->e1, st));
  if (n_obs > 5) {                                                               // :320-321
    rc = mark_general(ctx, m, f, obs, n_obs, st);
    if (rc != DDDMR_OK) return rc;
  }
  MK_LAUNCH(m, k_mk_finish, dim3((m->table + 255) / 256), dim3(256), 0, st, k, s, m->counters);
  if (timed) HIPCHK(ctx, hipEventRecord(m->e2, st));
  HIPCHK(ctx, hipMemcpyAsync(&out, m->counters, sizeof(out), hipMemcpyDeviceToHost, st));
  HIPCHK(ctx, hipStreamSynchronize(st));
  HIPCHK(ctx, hipGetLastError());
  ++m->updates_general;
  return DDDMR_OK;
}

// one update on the fused route: six launches, no copies (marking_fused.hip.h)
int update_fused(dddmr_rollout_ctx* ctx, MarkingState* m, const UpdateFrame& f, const float4* obs, uint32_t n_obs, bool timed,
                 MarkCounters& out) {
  hipStream_t st = m->cur;
  const MarkParams& k = f.k;
  MarkStore& s = m->store;
  if (!m->counters_clean) {
    MarkCounters zero{};
    zero.pool_used = m->pool_used_host;
    HIPCHK(ctx, hipMemcpyAsync(m->counters, &zero, sizeof(zero), hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemsetAsync(m->ticket, 0, 34 * sizeof(uint32_t), st));
  }
  m->counters_clean = false;                 // (until this update's last block has run)
  if (timed) HIPCHK(ctx, hipEventRecord(m->e0, st));
  int rc = store_maintenance(ctx, m, st);
  if (rc != DDDMR_OK) return rc;
  const bool mark = n_obs > 5;                                                   // :320-321
  const int cur = m->prev >= 0 ? 1 - m->prev : 0;
  GridBuf& gb = m->obs[cur];
  const PointGrid empty_grid = m->obs[0].g;
  const PointGrid prev_grid = m->prev >= 0 ? m->obs[m->prev].g : empty_grid;
  if (mark) {
    float lo[3], hi[3];
    const uint32_t cap = std::min(gb.cap_cells, m->fuse_cells);
    obs_grid_shape(m, f, cap, gb.g, lo, hi);
    gb.g.n = n_obs;
  }
  const uint32_t n_alive = m->n_alive_host;
  // ground cells of the window + inflation radius: the nodes the node-by-node dGraph updates look at
  SplatRange rg{};
  {
    float lo[3], hi[3];
    crop_box(m, f, lo, hi);
    const PointGrid& gg = m->ground.g;
    const float pad = (float)m->cfg.inflation_radius + 0.05f;
    rg.cx0 = host_cx(gg, lo[0] - pad); rg.cx1 = host_cx(gg, hi[0] + pad);
    rg.cy0 = host_cy(gg, lo[1] - pad); rg.cy1 = host_cy(gg, hi[1] + pad);
    rg.rows = (uint32_t)(rg.cy1 - rg.cy0 + 1) * (uint32_t)gg.nz;
    rg.segs = std::max(1u, (m->ground.max_row + 63u) / 64u);
    rg.delta = (int)std::floor(((float)m->cfg.inflation_radius + 1e-3f) * gg.inv_xy) + 1;
    rg.bands = (m->n_ground && (uint32_t)(rg.cy1 - rg.cy0 + 1) <= kBandMax) ? (uint32_t)(rg.cy1 - rg.cy0 + 1) : 0u;
  }
  FuseBufs fb{obs, m->parent, m->ds, m->gen, m->clear_list, m->unmark_pts,
              BandList{m->band_pts, m->band_cnt}, BandList{m->band_pts + (size_t)kBandMax * kBandCap, m->band_cnt + kBandMax}, rg,
              m->ticket, m->cell_count, m->keys_a, m->host_out_dev, m->grid_in_lds ? 1u : 0u, m->hi_rank};
  // Blocks that share one row segment's bands in the commit launch: a block takes every n_part-th 256-point chunk of the
  // 2 delta + 1 bands in reach, so more blocks than chunks only add blocks that read the band counts and leave (measured at
  // C5M, ~110 points per band, 9 bands in reach: n_part 48 / 24 / 16 / 12 / 8 -> mark phase 80 / 68 / 68 / 66 / 67 us).
  // Estimated from the observation size (at most one generator point per observation point, about half in practice).
  uint32_t n_part = m->splat_parts;
  if (n_part == 0) {
    const uint32_t per_band = std::max(1u, n_obs / 2u / std::max(1u, rg.bands));
    const uint32_t chunks = (2u * (uint32_t)rg.delta + 1u) * ((per_band + 255u) / 256u);
    n_part = std::min(48u, std::max(8u, chunks + chunks / 3u));
  }
  const uint32_t seg_groups = (rg.segs + 3u) / 4u;
  if (m->seq == 3 && std::getenv("DDDMR_DEBUG_GRID"))
    std::fprintf(stderr, "[dddmr] marking update: %u points, %u alive; window rows %u x %u segments, %u bands, delta %d -> %u blocks per row segment\n",
                 n_obs, m->n_alive_host, rg.rows, rg.segs, rg.bands, rg.delta, n_part);
  const uint32_t nb_band = rg.bands ? rg.rows * seg_groups * n_part : 0u;
  // (DDDMR_MKF_GRID=global only) cell counts of the observation grid
  const bool big = n_obs > kFuseRegObs;          // (the count launch's form of the grid only takes kFuseRegObs points)
  if (mark && !m->grid_in_lds && !big) MK_LAUNCH(m, k_mkf_count, dim3((n_obs + 255) / 256), dim3(256), 0, st, gb.g, fb);
  // grid launch: the observation grid (built in LDS by the first grid_parts workgroups) | every store slot: window + FOV test -> ray-test list
  {
    const uint32_t nb_grid = mark ? ((m->grid_in_lds || big) ? m->grid_parts : 1u) : 0u;
    if (big) MK_LAUNCH(m, k_mkf_grid_fov<true>, dim3(nb_grid + (m->table + 1023) / 1024), dim3(1024), 0, st, k, s, gb.g, fb, m->counters, nb_grid);
    else MK_LAUNCH(m, k_mkf_grid_fov<false>, dim3(nb_grid + (m->table + 1023) / 1024), dim3(1024), 0, st, k, s, gb.g, fb, m->counters, nb_grid);
  }
#ifdef DDDMR_PHASE_STAMPS
  if (const char* e = std::getenv("DDDMR_MKF_EXP")) { const int v = (std::atoi(e) & 128) ? 1 : 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(dddmr::g_mk_exp_noprobe), &v, sizeof(v)); }
#endif
  // clear launch: ray tests | union-find
  uint32_t nb_clear = (n_alive + 3) / 4, nb_cc = mark ? (n_obs * 4 + 255) / 256 : 0;
#ifdef DDDMR_PHASE_STAMPS
  // diagnostic build only (results are wrong with either): time the two halves of the launch apart
  if (const char* e = std::getenv("DDDMR_MKF_EXP")) { if (std::atoi(e) & 1) nb_cc = 0; if (std::atoi(e) & 2) nb_clear = 0; }
#endif
  if (nb_clear + nb_cc)
    MK_LAUNCH(m, k_mkf_clear_cc, dim3(nb_clear + nb_cc), dim3(256), 0, st, k, s, prev_grid, gb.g, m->ground.g, fb, m->counters, nb_clear);
  if (timed) HIPCHK(ctx, hipEventRecord(m->e1, st));
  uint32_t nb_un_groups = 0, nb_band_groups = 0;
  // seed launch: seeds (| removePCPtr of the cleared markings when it does not ride in the partition launch)
  {
    // (the removed markings of one update are a tenth of the new generator points: fewer, longer blocks per row)
    const uint32_t n_part_un = m->unmark_parts;
    uint32_t nb_roots = mark ? (n_obs + 255) / 256 : 0, nb_un = (n_alive && rg.bands) ? rg.rows * seg_groups * n_part_un : 0u, nb_walk = n_alive ? 64u : 0u;
#ifdef DDDMR_PHASE_STAMPS
    if (const char* e = std::getenv("DDDMR_MKF_EXP")) { if (std::atoi(e) & 32) nb_un = 0; if (std::atoi(e) & 64) nb_walk = 0; }
#endif
    if (mark && m->unmark_with_groups) { nb_un_groups = nb_un + nb_walk; nb_band_groups = nb_un; nb_un = 0; nb_walk = 0; }
    if (nb_roots + nb_un + nb_walk)
      MK_LAUNCH(m, k_mkf_roots_unmark, dim3(nb_roots + nb_un + nb_walk), dim3(256), 0, st, k, fb, m->cl, s, m->ground.g, m->counters, nb_roots,
                nb_un, seg_groups, n_part_un);
  }
  // partition launch: 64 partitions of the clusters | removePCPtr: ground node by ground node, point by point for the points that found no band
  if (mark)
    MK_LAUNCH(m, k_mkf_groups, dim3(kFuseParts + nb_un_groups), dim3(kPartThreads), kPartLdsBytes, st, k, fb, m->cl, s, m->ground.g, m->map.g, m->n_map,
              m->counters, nb_band_groups, seg_groups, m->unmark_parts);
  // commit launch: keepers -> pool | dGraph of the new generator points (node by node); the last block publishes the counters
  {
    uint32_t nb_commit = mark ? (n_obs + 255) / 256 : 0, nb_b = mark ? nb_band : 0u, nb_walk = mark ? 256u : 1u;
#ifdef DDDMR_PHASE_STAMPS
    if (const char* e = std::getenv("DDDMR_MKF_EXP")) { if (std::atoi(e) & 4) nb_commit = 0; if (std::atoi(e) & 8) nb_b = 0; if (std::atoi(e) & 16) nb_walk = 1; }
#endif
    MK_LAUNCH(m, k_mkf_commit_dgraph, dim3(nb_commit + nb_b + nb_walk), dim3(256), 0, st, k, fb, m->cl, s, m->ground.g, m->counters, nb_commit,
              nb_b, seg_groups, n_part);
  }
  if (timed) HIPCHK(ctx, hipEventRecord(m->e2, st));
  HIPCHK(ctx, hipStreamSynchronize(st));
  HIPCHK(ctx, hipGetLastError());
  out = *m->host_out;
  m->counters_clean = true;
  m->alive_list_stale = true;
  ++m->updates_fused;
  for (uint32_t v : out.n_cleared_shard) out.n_cleared += v;
  out.n_alive = n_alive - out.n_cleared + out.n_revived;
  out.n_clusters = n_obs;                                                        // extent of the per-cluster records (named by seed index)
  if (mark && !out.fallback) {
    m->prev = cur;                                                               // pcl_msg_gbl_ of this selfMark
    m->n_prev = n_obs;
  }
  if (!out.fallback) return DDDMR_OK;
  // A cluster beyond one partition workgroup, or voxel ranges beyond the 28-bit sort keys (points hundreds of metres
  // apart: only a cloud handed over with set_cloud can do that): the clear phase stands, nothing of the mark phase
  // has reached the store but keys and claims, and the mark phase is redone on the general route.
  --m->updates_fused;
  m->counters_clean = false;
  HIPCHK(ctx, hipMemsetAsync(s.owner, 0, (size_t)m->table * sizeof(unsigned long long), st));
  MarkCounters zero{};
  zero.pool_used = out.pool_used;
  HIPCHK(ctx, hipMemcpyAsync(m->counters, &zero, sizeof(zero), hipMemcpyHostToDevice, st));
  rc = mark_general(ctx, m, f, obs, n_obs, st);
  if (rc != DDDMR_OK) return rc;
  MK_LAUNCH(m, k_mk_finish, dim3((m->table + 255) / 256), dim3(256), 0, st, k, s, m->counters);
  if (timed) HIPCHK(ctx, hipEventRecord(m->e2, st));
  MarkCounters g{};
  HIPCHK(ctx, hipMemcpyAsync(&g, m->counters, sizeof(g), hipMemcpyDeviceToHost, st));
  HIPCHK(ctx, hipStreamSynchronize(st));
  HIPCHK(ctx, hipGetLastError());
  g.n_in_window = out.n_in_window; g.n_cleared = out.n_cleared; g.n_removed = out.n_removed; g.n_rehashed = out.n_rehashed;
  g.n_new_keys += out.n_new_keys;
  g.fallback = 1u;
  out = g;
  m->alive_list_stale = false;
  ++m->updates_general;
  return DDDMR_OK;
}
}  // namespace

extern "C" {

int dddmr_rollout_marking_update(dddmr_rollout_ctx* ctx, const double T_base_sensor[7], const double T_gbl_base[7],
                                 dddmr_marking_stats* stats) {
  if (!ctx || !T_base_sensor || !T_gbl_base) return DDDMR_ERR_BAD_ARG;
  std::lock_guard<std::mutex> tk(ctx->tick_mu);
  MarkingState* m = ctx->marking;
  if (!m) return fail(ctx, DDDMR_ERR_STATE, "marking_update before marking_create");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const dddmr_marking_config& c = m->cfg;
  // The observation = the context's published aggregate cloud (global frame), pinned like a tick pins it.
  // While a tick_begin is pending -- the reference runs the perception thread's doClear_then_Mark and the planner's
  // tick side by side -- the update runs on a stream of its own, next to the tick's kernels (the two touch disjoint
  // state), provided it reads the SAME observation the pending tick has pinned: a third pinned buffer would leave a
  // producer none to write into.
  const bool overlapped = ctx->pend.active;
  int cidx;
  struct Release { dddmr_rollout_ctx* c; ~Release() { if (c) release_cloud(c); } } release{nullptr};
  if (overlapped) {
    {
      std::lock_guard<std::mutex> lk(ctx->cloud_mu);
      if (ctx->busy < 0 || ctx->busy != ctx->front)
        return fail(ctx, DDDMR_ERR_STATE, "marking_update while a tick_begin is pending: a newer observation was published after tick_begin");
      cidx = ctx->busy;
    }
    if (!m->own_stream) HIPCHK(ctx, hipStreamCreateWithFlags(&m->own_stream, hipStreamNonBlocking));
    m->cur = m->own_stream;
    HIPCHK(ctx, hipStreamWaitEvent(m->cur, ctx->cloud_ready[cidx], 0));      // (the tick's stream has its own wait enqueued)
    ++m->updates_overlapped;
  } else {
    bool pending;
    cidx = pin_front(ctx, &pending);
    release.c = ctx;
    m->cur = ctx->stream;
    if (pending) {
      HIPCHK(ctx, hipStreamWaitEvent(m->cur, ctx->cloud_ready[cidx], 0));
      cloud_wait_done(ctx, cidx);
    }
  }
  const uint32_t n_obs = ctx->cloud_n[cidx];
  const float4* obs = ctx->cloud_dev[cidx];
  if (n_obs > m->max_obs) return fail(ctx, DDDMR_ERR_CAPACITY, "marking_update: observation of %u points, layer sized for %u", n_obs, m->max_obs);

  // ---- transforms (host, double): trans_gbl2s_af3_ = gbl2b * b2s (:236-237), its tf2 form (:238) ----
  UpdateFrame f{};
  MarkParams& k = f.k;
  double Rbs[9], Rs_e[9], ts[3];
  quat_to_rot(T_gbl_base, f.Rb);
  quat_to_rot(T_base_sensor, Rbs);
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) Rs_e[3 * i + j] = f.Rb[3 * i + 0] * Rbs[0 + j] + f.Rb[3 * i + 1] * Rbs[3 + j] + f.Rb[3 * i + 2] * Rbs[6 + j];
    ts[i] = f.Rb[3 * i + 0] * T_base_sensor[0] + f.Rb[3 * i + 1] * T_base_sensor[1] + f.Rb[3 * i + 2] * T_base_sensor[2] + T_gbl_base[i];
    f.t_gb[i] = T_gbl_base[i];
  }
  double qs[4];
  rot_to_quat(Rs_e, qs);
  tf2_set_rotation(qs, k.Rs);
  quat_rotate_z(qs, k.sn);
  for (int i = 0; i < 3; ++i) k.st[i] = ts[i];
  k.sd = -ts[0] * k.sn[0] - ts[1] * k.sn[1] - ts[2] * k.sn[2];
  {
    const double qb[4] = {T_gbl_base[3], T_gbl_base[4], T_gbl_base[5], T_gbl_base[6]};
    double nb[3];
    quat_rotate_z(qb, nb);
    k.mc[0] = (float)nb[0]; k.mc[1] = (float)nb[1]; k.mc[2] = (float)nb[2];
    const double d = -T_gbl_base[0] * nb[0] - T_gbl_base[1] * nb[1] - T_gbl_base[2] * nb[2];
    k.mc[3] = (float)d;
  }
  k.res = c.xy_resolution; k.hres = c.height_resolution; k.marking_height = c.marking_height; k.window = c.perception_window_size;
  k.fov_top = c.vertical_FOV_top; k.fov_bottom = c.vertical_FOV_bottom;
  k.ps = c.scan_effective_positive_start; k.pe = c.scan_effective_positive_end;
  k.ns = c.scan_effective_negative_start; k.ne = c.scan_effective_negative_end;
  k.ignore_ratio = c.segmentation_ignore_ratio; k.inscribed = c.inscribed_radius; k.inflation = c.inflation_radius;
  k.tol = (float)c.euclidean_cluster_extraction_tolerance;
  k.tol2 = static_cast<float>(c.euclidean_cluster_extraction_tolerance * c.euclidean_cluster_extraction_tolerance);
  k.min_cluster = c.euclidean_cluster_extraction_min_cluster_size;
  k.wx0 = (int)((T_gbl_base[0] - c.perception_window_size) / c.xy_resolution);      // :489-496
  k.wx1 = (int)((T_gbl_base[0] + c.perception_window_size) / c.xy_resolution);
  k.wy0 = (int)((T_gbl_base[1] - c.perception_window_size) / c.xy_resolution);
  k.wy1 = (int)((T_gbl_base[1] + c.perception_window_size) / c.xy_resolution);
  k.wz0 = (int)((T_gbl_base[2] - c.marking_height) / c.height_resolution);
  k.wz1 = (int)((T_gbl_base[2] + c.marking_height) / c.height_resolution);
  k.n_obs = n_obs;
  k.n_prev = m->prev >= 0 ? m->n_prev : 0;
  k.pad = 1e-4f;
  if (const char* e = std::getenv("DDDMR_MK_PAD")) k.pad = (float)std::atof(e);      // (diagnosis)
  k.table_mask = m->table - 1;
  k.pool_cap = m->pool_cap;
  k.n_ground = m->n_ground;
  k.n_alive_prev = m->n_alive_host;
  k.seq = ++m->seq;
  if (k.seq == 0) k.seq = m->seq = 1;

  // HIP events serialise the queue around them: the update is timed like the tick (DDDMR_TIMING / _EVERY)
  const bool timed = stats && ctx->timing >= 1 && (m->seq % (uint32_t)ctx->timing_every) == 0;
  MarkCounters out{};
  m->launches_last = 0;
  const bool fused = m->route != 0 && n_obs <= kFuseMaxObs;
  if (m->route == 1 && !fused)
    return fail(ctx, DDDMR_ERR_CAPACITY, "marking_update: DDDMR_MARKING_ROUTE=fused, observation of %u points > %u", n_obs, kFuseMaxObs);
  const int rc = fused ? update_fused(ctx, m, f, obs, n_obs, timed, out) : update_general(ctx, m, f, obs, n_obs, timed, out);
  if (rc != DDDMR_OK) return rc;
  m->pool_used_host = out.pool_used;
  m->n_alive_host = out.n_alive;
  m->keys_used_host += out.n_new_keys;
  if (out.n_dup > 0 && !out.overflow) {
    const int rt = marking_fix_ties(ctx, m, k, m->store, out);
    if (rt != DDDMR_OK) return rt;
  }
  if (timed) {
    HIPCHK(ctx, hipEventElapsedTime(&m->last_clear_ms, m->e0, m->e1));
    HIPCHK(ctx, hipEventElapsedTime(&m->last_mark_ms, m->e1, m->e2));
  }
  if (stats) {
    stats->n_observation = n_obs > 5 ? n_obs : 0;
    stats->n_clusters = out.n_clusters_kept;
    stats->n_marked = out.n_marked;
    stats->n_in_window = out.n_in_window;
    stats->n_cleared = out.n_cleared;
    stats->n_alive = out.n_alive;
    stats->clear_ms = m->last_clear_ms;       // the latest timed update's
    stats->mark_ms = m->last_mark_ms;
  }
  if (out.overflow)
    return fail(ctx, DDDMR_ERR_CAPACITY, "marking_update: capacity flag %u (1: max_markings, 2: max_cluster_points, 4: a cluster more than 3.2 km (51 m in z) from the window)", out.overflow);
  return DDDMR_OK;
}

int dddmr_rollout_marking_route_counts(dddmr_rollout_ctx* ctx, uint32_t* updates_fused, uint32_t* updates_general,
                                       uint32_t* launches_last_update) {
  if (!ctx) return DDDMR_ERR_BAD_ARG;
  std::lock_guard<std::mutex> tk(ctx->tick_mu);
  MarkingState* m = ctx->marking;
  if (!m) return fail(ctx, DDDMR_ERR_STATE, "marking_route_counts before marking_create");
  if (updates_fused) *updates_fused = m->updates_fused;
  if (updates_general) *updates_general = m->updates_general;
  if (launches_last_update) *launches_last_update = m->launches_last;
  return DDDMR_OK;
}

#ifdef DDDMR_PHASE_STAMPS
// diagnostic build only: phase stamps of the last fused update's two single-workgroup blocks
int dddmr_rollout_diag_mkstamps(unsigned long long* out, size_t n_words) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(dddmr::g_mk_stamps), n_words * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
int dddmr_rollout_diag_mkclear(unsigned long long* out) {
#ifdef DDDMR_CLEAR_CYCLES
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(dddmr::g_mk_clear_cyc), 8 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
#else
  for (int i = 0; i < 8; ++i) out[i] = 0;
  return 0;
#endif
}
#endif

int dddmr_rollout_marking_get_voxels(dddmr_rollout_ctx* ctx, int32_t* xyz_out, size_t capacity, size_t* n) {
  if (!ctx || !n) return DDDMR_ERR_BAD_ARG;
  std::lock_guard<std::mutex> tk(ctx->tick_mu);
  MarkingState* m = ctx->marking;
  if (!m) return fail(ctx, DDDMR_ERR_STATE, "marking_get_voxels before marking_create");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  std::vector<unsigned long long> keys(m->table);
  std::vector<uint32_t> alive(m->table);
  HIPCHK(ctx, hipMemcpy(keys.data(), m->store.keys, keys.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  HIPCHK(ctx, hipMemcpy(alive.data(), m->store.alive, alive.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
  size_t cnt = 0;
  for (size_t i = 0; i < keys.size(); ++i) {
    if (!alive[i] || !keys[i]) continue;
    if (xyz_out) {
      if (cnt >= capacity) return fail(ctx, DDDMR_ERR_CAPACITY, "marking_get_voxels: capacity %zu too small", capacity);
      int x, y, z;
      voxel_unkey(keys[i], &x, &y, &z);
      xyz_out[3 * cnt] = x; xyz_out[3 * cnt + 1] = y; xyz_out[3 * cnt + 2] = z;
    }
    ++cnt;
  }
  *n = cnt;
  return DDDMR_OK;
}

// The generator points of every alive marking (the cluster projected on the robot's ground plane, 0.1 m VoxelGrid): what
// Marking::computeMinDistanceFromObstacle2GroundNodes searches the ground with (cluster_marking.cpp:54-64).  Debug /
// visualisation (the reference publishes its markings as a cloud); order: by store slot.
int dddmr_rollout_marking_get_points(dddmr_rollout_ctx* ctx, float* xyz_out, int32_t* voxel_out, size_t capacity, size_t* n) {
  if (!ctx || !n) return DDDMR_ERR_BAD_ARG;
  std::lock_guard<std::mutex> tk(ctx->tick_mu);
  MarkingState* m = ctx->marking;
  if (!m) return fail(ctx, DDDMR_ERR_STATE, "marking_get_points before marking_create");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  std::vector<uint32_t> alive(m->table), ofs(m->table), cnt(m->table);
  std::vector<unsigned long long> keys(m->table);
  HIPCHK(ctx, hipMemcpy(keys.data(), m->store.keys, keys.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  HIPCHK(ctx, hipMemcpy(alive.data(), m->store.alive, alive.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
  HIPCHK(ctx, hipMemcpy(ofs.data(), m->store.pts_ofs, ofs.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
  HIPCHK(ctx, hipMemcpy(cnt.data(), m->store.pts_n, cnt.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
  size_t total = 0;
  for (size_t i = 0; i < alive.size(); ++i)
    if (alive[i]) total += cnt[i];
  *n = total;
  if (!xyz_out) return DDDMR_OK;
  if (total > capacity) return fail(ctx, DDDMR_ERR_CAPACITY, "marking_get_points: capacity %zu < %zu", capacity, total);
  std::vector<float4> pool(m->pool_used_host);
  if (!pool.empty()) HIPCHK(ctx, hipMemcpy(pool.data(), m->store.pool, pool.size() * sizeof(float4), hipMemcpyDeviceToHost));
  size_t at = 0;
  for (size_t i = 0; i < alive.size(); ++i) {
    if (!alive[i]) continue;
    for (uint32_t j = 0; j < cnt[i]; ++j, ++at) {
      if ((size_t)ofs[i] + j >= pool.size()) return fail(ctx, DDDMR_ERR_STATE, "marking_get_points: slot %zu points past the pool", i);
      const float4 p = pool[(size_t)ofs[i] + j];
      xyz_out[3 * at] = p.x; xyz_out[3 * at + 1] = p.y; xyz_out[3 * at + 2] = p.z;
      if (voxel_out) voxel_unkey(keys[i], &voxel_out[3 * at], &voxel_out[3 * at + 1], &voxel_out[3 * at + 2]);
    }
  }
  return DDDMR_OK;
}

int dddmr_rollout_marking_get_dgraph(dddmr_rollout_ctx* ctx, double* values_out, size_t capacity) {
  if (!ctx || !values_out) return DDDMR_ERR_BAD_ARG;
  std::lock_guard<std::mutex> tk(ctx->tick_mu);
  MarkingState* m = ctx->marking;
  if (!m) return fail(ctx, DDDMR_ERR_STATE, "marking_get_dgraph before marking_create");
  if (capacity < (size_t)m->n_ground + 1) return fail(ctx, DDDMR_ERR_CAPACITY, "marking_get_dgraph: capacity %zu < %u", capacity, m->n_ground + 1);
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipMemcpy(values_out, m->store.dgraph, ((size_t)m->n_ground + 1) * sizeof(double), hipMemcpyDeviceToHost));
  return DDDMR_OK;
}

int dddmr_rollout_marking_get_lethal(dddmr_rollout_ctx* ctx, uint8_t* flags_out, size_t capacity) {
  if (!ctx || !flags_out) return DDDMR_ERR_BAD_ARG;
  std::lock_guard<std::mutex> tk(ctx->tick_mu);
  MarkingState* m = ctx->marking;
  if (!m) return fail(ctx, DDDMR_ERR_STATE, "marking_get_lethal before marking_create");
  if (capacity < (size_t)m->n_ground + 1) return fail(ctx, DDDMR_ERR_CAPACITY, "marking_get_lethal: capacity %zu < %u", capacity, m->n_ground + 1);
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipMemcpy(flags_out, m->store.lethal, (size_t)m->n_ground + 1, hipMemcpyDeviceToHost));
  return DDDMR_OK;
}

}  // extern "C"
