// gjx_comm.hip -- the collectives of the sharded smc step issued from the C side, and
// the step's front end as ONE host call.
//
// The reference has no multi-device code (src/genjax/core.py:1304-1317 is dead plumbing): this is new work, whose
// semantics are resample()'s (src/genjax/inference/smc.py:603-652) over a particle axis sharded by contiguous index
// ranges.  Between extend and the move launch of a sharded step sit three small all_gathers (the shard maxima, the
// tile sums of the normaliser, the CDF message); driven from Python each one costs a host round trip per rank
// (~11 us of idle GPU per collective, profiles/r02_sharded_step_timeline_after.txt).  Here the kernels and the
// collectives are enqueued back to back on the caller's stream by one call.
//
// RCCL is bound at run time (dlopen of librccl.so.1: inside a PyTorch process that is the instance torch already
// loaded, so one process never holds two copies of the library); a gjx_comm of one rank needs no RCCL at all (its
// all_gather is a device-to-device copy).  torch.distributed is only the bootstrap: rank 0's unique id travels over
// the existing process group (genjax_amd/parallel.py).
#include <dlfcn.h>
#include <pthread.h>
#include <rccl/rccl.h>

#include "gjx_common.h"

struct gjx_comm {
  int rank, world;
  ncclComm_t comm;
};

namespace {
struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi g_rccl;
pthread_mutex_t g_rccl_lock = PTHREAD_MUTEX_INITIALIZER;

int32_t rccl_api(RcclApi** out) {
  pthread_mutex_lock(&g_rccl_lock);
  if (!g_rccl.lib) {
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (h) {
      RcclApi a;
      a.lib = h;
      a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(h, "ncclGetUniqueId");
      a.CommInitRank = (decltype(a.CommInitRank))dlsym(h, "ncclCommInitRank");
      a.CommDestroy = (decltype(a.CommDestroy))dlsym(h, "ncclCommDestroy");
      a.AllGather = (decltype(a.AllGather))dlsym(h, "ncclAllGather");
      a.GetErrorString = (decltype(a.GetErrorString))dlsym(h, "ncclGetErrorString");
      if (a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllGather && a.GetErrorString) g_rccl = a;
    }
  }
  pthread_mutex_unlock(&g_rccl_lock);
  GJX_REQUIRE(g_rccl.lib != nullptr, GJX_EHIP, "RCCL is not available (dlopen librccl.so.1: %s)", dlerror());
  *out = &g_rccl;
  return GJX_OK;
}
}  // namespace

#define GJX_CHECK_NCCL(api, expr)                                                                        \
  do {                                                                                                   \
    ncclResult_t _r = (expr);                                                                            \
    if (_r != ncclSuccess) {                                                                             \
      gjx_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr, (api)->GetErrorString(_r));        \
      return GJX_EHIP;                                                                                   \
    }                                                                                                    \
  } while (0)

// 128 bytes that rank 0 creates and every rank of the group passes to gjx_comm_create.
GJX_API int32_t gjx_comm_unique_id(uint8_t* out128) {
  GJX_REQUIRE(out128 != nullptr, GJX_EINVAL, "gjx_comm_unique_id: out is null");
  RcclApi* api = nullptr;
  int32_t rc = rccl_api(&api);
  if (rc != GJX_OK) return rc;
  ncclUniqueId id;
  static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
  GJX_CHECK_NCCL(api, api->GetUniqueId(&id));
  memcpy(out128, &id, sizeof(id));
  return GJX_OK;
}

// One communicator per (process, GPU): the calling thread's current device is the rank's GPU.  world == 1 needs no
// id and then no RCCL (id128 null: all_gather is a copy).  Collective over the group: every rank calls it with the
// same id.
GJX_API int32_t gjx_comm_create(const uint8_t* id128, int32_t rank, int32_t world, gjx_comm** out) {
  GJX_REQUIRE(out != nullptr, GJX_EINVAL, "gjx_comm_create: out is null");
  GJX_REQUIRE(world >= 1 && world <= GJX_MAX_RANKS && rank >= 0 && rank < world, GJX_EINVAL,
              "gjx_comm_create: rank %d of %d", rank, world);
  gjx_comm* c = new gjx_comm();
  c->rank = rank;
  c->world = world;
  c->comm = nullptr;
  if (world > 1 && id128 == nullptr) {
    delete c;
    gjx_set_error("gjx_comm_create: %d ranks need the unique id of gjx_comm_unique_id", world);
    return GJX_EINVAL;
  }
  if (id128 != nullptr) {   // (one rank WITH an id: a real one-rank RCCL communicator -- what a one-GPU box can test)
    RcclApi* api = nullptr;
    int32_t rc = rccl_api(&api);
    if (rc != GJX_OK) {
      delete c;
      return rc;
    }
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclResult_t r = api->CommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) {
      gjx_set_error("gjx_comm_create: ncclCommInitRank failed: %s", api->GetErrorString(r));
      delete c;
      return GJX_EHIP;
    }
  }
  *out = c;
  return GJX_OK;
}

GJX_API int32_t gjx_comm_destroy(gjx_comm* c) {
  if (!c) return GJX_OK;
  if (c->comm) {
    RcclApi* api = nullptr;
    if (rccl_api(&api) == GJX_OK) (void)api->CommDestroy(c->comm);
  }
  delete c;
  return GJX_OK;
}

// recv[r * bytes .. (r + 1) * bytes) = rank r's send, on `stream` (in place when send == recv + rank * bytes).
GJX_API int32_t gjx_comm_all_gather(gjx_comm* c, const void* d_send, void* d_recv, int64_t bytes_per_rank,
                                    void* stream) {
  GJX_REQUIRE(c && d_send && d_recv && bytes_per_rank > 0, GJX_EINVAL, "gjx_comm_all_gather: bad argument");
  hipStream_t s = (hipStream_t)stream;
  if (c->comm == nullptr) {   // one rank, no RCCL
    if (d_send != d_recv)
      GJX_CHECK_HIP(hipMemcpyAsync(d_recv, d_send, (size_t)bytes_per_rank, hipMemcpyDeviceToDevice, s));
    return GJX_OK;
  }
  RcclApi* api = nullptr;
  int32_t rc = rccl_api(&api);
  if (rc != GJX_OK) return rc;
  if (bytes_per_rank % 4 == 0)
    GJX_CHECK_NCCL(api, api->AllGather(d_send, d_recv, (size_t)(bytes_per_rank / 4), ncclFloat32, c->comm, s));
  else
    GJX_CHECK_NCCL(api, api->AllGather(d_send, d_recv, (size_t)bytes_per_rank, ncclUint8, c->comm, s));
  return GJX_OK;
}

// The front end of one sharded mh step in one call (what genjax_amd/parallel.py otherwise drives call by call):
//   extend (+ this shard's maximum)            -> all_gather 1 float per rank
//   tile sums relative to the global maximum   -> all_gather 2 floats per 2048-tile
//   fold -> {max, lse, ESS} of the GLOBAL weight vector (the shard-maximum cell goes back to -inf)
//   CDF message                                -> all_gather
//   CDF slice + strips + coarse table, owner-side search, the split {p0, p1, go} with the lax.cond decided on the
//   device.
// With move_keys the move launch of the own-sourced positions (gjx_shard_moves_main) follows in the same call, so that
// nothing waits for the host between the search and the moves; ev_split (optional event) is recorded in between, where
// the step's side work forks: counts / pack / exchange / gjx_shard_moves_fixup run beside the move launch.
// *b->max1 must hold -inf on entry (the caller initialises it so; this call leaves it so).
GJX_API int32_t gjx_shard_front(gjx_ctx* ctx, gjx_comm* comm, const gjx_loc_world* world, const uint32_t keys[3][2],
                                int32_t is_initial, const float* obs, float u, const gjx_shard_bufs* b,
                                int32_t n_moves, int64_t* d_accept_counts, const uint32_t (*move_keys)[4][2],
                                int32_t sel_mask, float* const* d_cols_alt, void* ev_split, void* ev_moves_begin,
                                void* ev_moves_end, void* stream) {
  GJX_REQUIRE(ctx && comm && world && keys && obs && b, GJX_EINVAL, "gjx_shard_front: null argument");
  GJX_REQUIRE(b->n > 0 && b->n % GJX_SCAN_TILE == 0 && b->n_ranks == comm->world && b->rank == comm->rank &&
                  b->n_global == b->n * (int64_t)b->n_ranks,
              GJX_EINVAL, "gjx_shard_front: buffers of rank %d / %d with n = %lld, communicator of rank %d / %d", b->rank,
              b->n_ranks, (long long)b->n, comm->rank, comm->world);
  for (int c = 0; c < 6; ++c) GJX_REQUIRE(b->cols[c] != nullptr, GJX_EINVAL, "gjx_shard_front: column %d is null", c);
  GJX_REQUIRE(b->logw && b->max1 && b->max_all && b->tiles && b->tiles_all && b->gstats && b->msg && b->msg_all &&
                  b->probe_nodes && b->probe_mids && b->all_totals && b->scanned && b->cdf_ext && b->table && b->range &&
                  b->cand_idx && b->tile_counts && b->counts && b->split,
              GJX_EINVAL, "gjx_shard_front: a buffer is null");
  const int64_t n = b->n, nt = n / GJX_SCAN_TILE;
  const int G = b->n_ranks;
  int32_t rc = gjx_loc_shard_extend_cell(ctx, world, keys, is_initial, (uint64_t)b->rank * (uint64_t)n, n, obs,
                                         is_initial ? nullptr : b->cols[0], is_initial ? nullptr : b->cols[1],
                                         is_initial ? nullptr : b->cols[2], b->logw, b->cols[0], b->cols[1], b->cols[2],
                                         b->cols[3], b->cols[4], b->cols[5], b->logw, b->max1, b->ray_scores,
                                         b->scores_stride, stream);
  if (rc != GJX_OK) return rc;
  rc = gjx_comm_all_gather(comm, b->max1, b->max_all, sizeof(float), stream);
  if (rc != GJX_OK) return rc;
  rc = gjx_shard_tile_sums(b->logw, n, b->max_all, G, b->tiles, stream);
  if (rc != GJX_OK) return rc;
  rc = gjx_comm_all_gather(comm, b->tiles, b->tiles_all, (int64_t)sizeof(float) * 2 * nt, stream);
  if (rc != GJX_OK) return rc;
  rc = gjx_shard_stats_fold_reset(b->tiles_all, nt * G, b->max_all, G, b->gstats, b->max1, stream);
  if (rc != GJX_OK) return rc;
  rc = gjx_shard_cdf_begin(b->logw, b->gstats + 1, n, b->rank, b->probe_nodes + (int64_t)b->rank * b->p_cap,
                           b->probe_mids + (int64_t)b->rank * b->p_cap, b->p_cap, b->msg, stream);
  if (rc != GJX_OK) return rc;
  rc = gjx_comm_all_gather(comm, b->msg, b->msg_all, (int64_t)sizeof(float) * b->msg_len, stream);
  if (rc != GJX_OK) return rc;
  rc = gjx_shard_cdf_finish(ctx, b->logw, b->gstats + 1, n, b->msg_all, G, b->rank, b->probe_nodes, b->probe_mids,
                            b->p_cap, b->all_totals, b->scanned, b->cdf_ext, b->lc, b->table, stream);
  if (rc != GJX_OK) return rc;
  // (the row counts are not on the way to the move launch: gjx_shard_count runs with the side work of the step)
  rc = gjx_shard_search_only(b->table, b->lc, b->cdf_ext, G, b->rank, n, u, b->cand_cap, b->range, b->cand_idx,
                             b->counts, stream);
  if (rc != GJX_OK) return rc;
  rc = gjx_shard_split(b->cand_idx, b->range, b->rank, n, b->gstats, b->n_global, b->split, d_accept_counts, n_moves,
                       stream);
  if (rc != GJX_OK) return rc;
  // the side work of the step (row counts, pack, read-back) forks here: it needs the search and the split, not the moves
  if (ev_split) GJX_CHECK_HIP(hipEventRecord((hipEvent_t)ev_split, (hipStream_t)stream));
  if (move_keys == nullptr) return GJX_OK;
  GJX_REQUIRE(d_cols_alt != nullptr && n_moves > 0, GJX_EINVAL, "gjx_shard_front: the moves need the alternate columns");
  return gjx_shard_moves_main(ctx, world, b->cand_idx, b->range, b->rank, n, b->split, move_keys, n_moves, sel_mask,
                              is_initial, obs, (float* const*)b->cols, d_cols_alt, b->logw, b->ray_scores,
                              b->scores_stride, d_accept_counts, ev_moves_begin, ev_moves_end, stream);
}
