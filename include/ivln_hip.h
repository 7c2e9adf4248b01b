/*
 * ivln_hip.h - C ABI of libivln_hip.so, the MI355X (gfx950) implementation of the IVLN-CE MapCMA
 * hot path.  Plain pointers and sizes only: every `void*`/typed pointer below is a DEVICE pointer
 * owned by the caller (contiguous, layout documented per call) unless marked "host"; `stream` is a
 * hipStream_t passed as void*.  All entry points return 0 on success or a negative IVLN_E_* code
 * (never throw, never allocate persistent memory outside *_create/_destroy handles).  Kernels are
 * enqueued on `stream` and the call returns immediately.
 *
 * The reference (jacobkrantz/IVLN-CE) is pure Python/torch and has no FFI: each entry point cites
 * the reference function whose arithmetic it replaces (paths relative to the reference root).  The
 * Python plugin classes in ivln-ce_amd/ bind these with ctypes.
 */
#ifndef IVLN_HIP_H
#define IVLN_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define IVLN_OK 0
#define IVLN_E_INVALID -1   /* bad argument */
#define IVLN_E_HIP -2       /* a HIP runtime call failed */
#define IVLN_E_KEYSPACE -3  /* mapper: keep-highest key exceeded the dense table capacity */
#define IVLN_E_CAPACITY -4  /* mapper: world cloud exceeded its capacity */
#define IVLN_E_UNSUPPORTED -5

const char* ivln_strerror(int code);
int ivln_version(void);

/* ------------------------------------------------------------------------------------------
 * Egocentric semantic mapper (replaces MappingModule.forward,
 * ivlnce_baselines/common/mapping_module/mapper.py:904-944 and everything it calls:
 * projector/core.py:117-230, mapper.py:236-266, 387-474, 477-648, 807-901).
 * ------------------------------------------------------------------------------------------ */
typedef struct ivln_mapper ivln_mapper;

/* B_max envs, depth H x W, vertical fov (radians), map size in metres and resolution
 * (obs_transforms.py:154-175 / setup_mapping_module.py:13-59).  world_capacity = max points in the
 * world cloud (0 -> default), table_cells = dense keep-highest table capacity (0 -> default). */
int ivln_mapper_create(int B_max, int H, int W, double vfov_rad, double height_m, double width_m,
                       double res_m, int64_t world_capacity, int64_t table_cells, ivln_mapper** out);
int ivln_mapper_destroy(ivln_mapper* m);
/* Forget the world cloud (new process / new eval). */
int ivln_mapper_reset(ivln_mapper* m, void* stream);

/* core.py:6-37 (_transform3D with elevation + pi, mapper.py:132-138) and mapper.py:38-48
 * (rotate_around_y_matrix(-heading)): pose f32 (B,3), orientation f64 (B,2) [elev, heading] ->
 * T f32 (B,4,4), rot f32 (B,3,3).  fp64 sin/cos on device, rounded to fp32. */
int ivln_mapper_frames(const float* pose, const double* orientation, int B, float* T, float* rot,
                       void* stream);

/* One mapper step.  depth f32 (B,H,W) in [0,1]; labels u8 (B,H,W) (semantic12 or RedNet argmax);
 * T f32 (B,4,4); pose f32 (B,3); rot f32 (B,3,3); not_done u8 (B) -> occupancy_map, semantic_map
 * u8 (B,rows,cols), both fully rewritten. */
int ivln_mapper_step(ivln_mapper* m, const float* depth, const uint8_t* labels, const float* T,
                     const float* pose, const float* rot, const uint8_t* not_done, int B,
                     uint8_t* occ_out, uint8_t* sem_out, void* stream);

/* Known-map mode (mapper.py:851-881): begin = clear finished/paused envs; load = append the
 * pre-built cloud of env b (xyz f32 (n,3), sem u8 (n), device); raster = maps from the cloud. */
int ivln_mapper_known_begin(ivln_mapper* m, const uint8_t* not_done, int B, void* stream);
int ivln_mapper_load_known(ivln_mapper* m, int b, const float* xyz, const uint8_t* sem, int64_t n,
                           void* stream);
int ivln_mapper_known_raster(ivln_mapper* m, const float* pose, const float* rot, int B,
                             uint8_t* occ_out, uint8_t* sem_out, void* stream);

/* Synchronises `stream` and returns the sticky device status (IVLN_OK / IVLN_E_KEYSPACE /
 * IVLN_E_CAPACITY); *world_n (host, may be NULL) receives the world cloud size. */
int ivln_mapper_status(ivln_mapper* m, int64_t* world_n, void* stream);
/* Debug/test export of the world cloud (unordered): xyz f32 (max_n,3), meta u32 (max_n) =
 * batch<<8|label, rank i64 (max_n) = position key of the reference's ordering.  Synchronises. */
int ivln_mapper_world_export(ivln_mapper* m, float* xyz, uint32_t* meta, int64_t* rank,
                             int64_t max_n, int64_t* n_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif
