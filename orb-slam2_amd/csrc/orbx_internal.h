// orbx_internal.h — shared host/device definitions of liborbx (not part of the public ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "../../include/orbx.h"

#define ORBX_MAX_LEVELS 16
#define ORBX_EDGE 19          // EDGE_THRESHOLD, reference src/ORBextractor.cc:74
#define ORBX_MIN_BORDER 16    // EDGE_THRESHOLD-3, src/ORBextractor.cc:934
#define ORBX_TILE_PITCH 80    // LDS pitch of a FAST cell tile when a cell is wider than 38 px (cell side <= 59 + 6 halo + 1): 3 rows per direct load
#define ORBX_SCORE_PITCH 64   // LDS pitch of a FAST cell score tile (detect side <= 59 + 2)
#define ORBX_MAX_DIM 4096     // packed candidate = x | y<<12 | score<<24
#define ORBX_NODE_BITS 14     // quadtree node id bits inside the per-point label
#define ORBX_CAND_PRIM 16      // a cell's first candidates live in a dense 64-byte record (cells contiguous); only the overflow uses its big slot block
#ifndef ORBX_TREE_REG_PTS
#define ORBX_TREE_REG_PTS 3072 // k_tree: a level with at most this many candidates keeps them in registers (12 per thread of 256)
#endif
#define ORBX_TREE_REG_PTS_BIG 4096 // the same for the 1024-thread form (4 per thread)
#ifndef ORBX_TREE_OVER_PTS
#define ORBX_TREE_OVER_PTS 1024 // register form: points of a level beyond the register capacity that an LDS overflow array takes
#endif
#define ORBX_FAST_LIST_CAP 512 // k_fast: pretest candidates listed per round (u16 each); denser cells take several rounds

// Geometry of one pyramid level for one image size (host computes, device reads).
struct LevelGeom {
    int w, h, pitch;        // pitch in bytes of the stored level (levels >= 1)
    long long pyr_off;      // byte offset of the level inside one image's pyramid block (levels >= 1)
    int n_cols, n_rows, w_cell, h_cell; // FAST cell grid, src/ORBextractor.cc:944-951
    int cell_base, n_cells; // index range of this level's cells in the per-image cell arrays
    int cand_cap;           // candidate slots per cell
    long long cand_off;     // first slot (u32 units) of this level in the per-image candidate block
    int quota;              // mnFeaturesPerLevel
    int n_ini; float hx;    // quadtree roots, src/ORBextractor.cc:627-628
    int tree_w, tree_h;     // maxBorder - minBorder
    int node_cap;           // max simultaneous quadtree leaves (+slack)
    int kp_off, kp_cap;     // slot range in the per-image per-level keypoint staging
    float scale;            // mvScaleFactor[level]
    int patch_size;         // (int)(31*scale)
    int tab_x, tab_y;       // offsets (int16 units) of the resize tables [ofs|c0|c1] x 3*w / 3*h
    int tab_tx, tab_ty;     // offsets of k_resize's per-tile-column / per-tile-row records
    int resize_lds;         // 1: k_resize's LDS tile fits this level's scale, 0: k_resize_direct, 2: exact 2x -> area average (k_resize_direct)
};

struct Geom {
    int nlevels, w, h;
    int total_cells;        // per image
    long long cand_total;   // per image, u32 units
    int kp_total;           // per image staging slots
    long long pyr_bytes;    // per image, levels >= 1
    int max_cells_level;    // max n_cells over levels
    int max_node_cap;
    int fast_lds_sc, fast_lds_list, fast_lds_bm, fast_bm_rows, fast_lds_bytes; // LDS carve of k_fast
    int fast_small;         // 1: k_fast<48,40> (every cell <= 38 px wide), 0: k_fast<80,64>
    int fast_list_cap_big, fast_lds_bm_big, fast_lds_bytes_big;   // the carve with a candidate list that holds a whole cell (several waves per cell)
    int fast2_ok, total_pairs;   // k_fast2 (a wave per pair of adjacent cells) is usable for this geometry; pair records per image
    int fast2_first[2], fast2_count[2];   // two launches: levels whose detect areas have at most 32 rows, then the taller ones
    int fast2_lds_sc[2], fast2_lds_list[2], fast2_lds_bm[2], fast2_bm_rows[2], fast2_lds_bytes[2];
    LevelGeom lv[ORBX_MAX_LEVELS];
};

// Per-cell record of k_fast (same for every image): one 32-byte scalar load replaces the level search
// and the cell-grid arithmetic (a chain of dependent scalar loads per wave).
struct CellRec {
    short level, skip;
    short ini_x, ini_y, tw, th;  // cell rectangle incl. 3-px halo, level coordinates
    int pitch;                   // level pitch (levels >= 1; level 0 uses the caller's pitch)
    int cand_cap;
    unsigned gpr_magic;          // 0xFFFFFFFF / gpr + 1, gpr = dword groups per detect row = (tw - 6 + 3) >> 2: multiply-high division by gpr
    long long pyr_off;           // byte offset of the level in one image's pyramid block
    long long cand_slot;         // first candidate slot (u32 units) of this cell in one image's block
};
static_assert(sizeof(CellRec) == 40, "CellRec layout");

struct ProfEvent { hipEvent_t a, b; int stage; bool owns_a; };

// One frame in flight of the pipelined host-pointer stereo path (orbx_extract_stereo_submit / _wait)
struct PipeSlot {
    uint8_t *h_in, *d_in; size_t in_cap;     // both eyes, pinned host staging and device level 0
    uint8_t *h_in_dev, *h_out_dev;           // the addresses a kernel reads / writes the two pinned buffers at
    uint8_t *h_out; size_t h_out_cap;        // pinned: [n0 n1 flag | keypoints x2 | descriptors x2 | uRight | depth]
    uint8_t *d_out;                          // one device block in the layout of h_out (a single download per frame)
    void *d_kps, *d_desc, *d_n; float *d_ur, *d_z; int out_cap;   // views into d_out
    hipEvent_t ev_h2d, ev_done, ev_d2h;      // input landed / kernels finished / results landed in h_out
    int cap, ticket, eyes; bool busy;
    struct orbx_extractor *lane;             // the kernel lane (the handle or its shadow) that ran the slot's frame
};
#define ORBX_PIPE_DEPTH 4

struct orbx_extractor {
    int device;
    int nfeatures, nlevels, ini_th, min_th;
    double scale_factor;
    int max_w, max_h, max_batch;
    float sf[ORBX_MAX_LEVELS], isf[ORBX_MAX_LEVELS], sig2[ORBX_MAX_LEVELS], isig2[ORBX_MAX_LEVELS];
    int quota[ORBX_MAX_LEVELS];
    int umax[16];
    int cv_profile, gauss[4];        // OpenCV generation whose GaussianBlur taps k_desc filters with (orbx_extractor_set_cv_profile)

    hipStream_t stream;
    hipStream_t last_launch_stream;  // stream of the most recent kernel launches on this handle's workspaces (caller's or `stream`)
    hipEvent_t ev_switch;            // orders a launch on a new stream behind the work of the previous one (orbx_use_stream)
    // geometry of the current image size
    Geom geom;           // host copy (geom.w == 0: none yet)
    Geom *d_geom;        // device copy
    int16_t *d_tabs; size_t tabs_cap;      // resize tables
    CellRec *d_cells; size_t cells_cap;    // per-cell records of k_fast
    void *d_pairs; size_t pairs_cap;       // per-pair records of k_fast2 (PairRec, orbx_extract.hip)
    int last_fast_form;                    // debug: 1 = k_fast, 2 = k_fast2 ran in the most recent extraction
    int fast_pair;                         // ORBX_FAST_PAIR: -1 = k_fast2 for batches, 0 = never, 1 = always
    // workspace (sized for max_w x max_h x max_batch)
    uint8_t *d_pyr; size_t pyr_cap;        // levels >= 1, all images
    uint8_t *d_stage_in; size_t stage_in_cap; // host-API input staging (level 0)
    int *d_cell_cnt; size_t cell_cnt_cap;
    uint32_t *d_cand; size_t cand_cap;
    uint32_t *d_cand_prim; size_t cand_prim_cap;   // [max_batch][total_cells][ORBX_CAND_PRIM]
    uint32_t *d_tree_pts; uint16_t *d_tree_nid; size_t tree_cap; // overflow scratch of the quadtree
    unsigned char *d_tree_tab; size_t tree_tab_cap; // quadtree node tables of configurations whose tables exceed the LDS (else unused)
    int *d_lvl_cnt;                        // [max_batch][nlevels]
    uint32_t *d_lvl_kp; size_t lvl_kp_cap; // [max_batch][kp_total]
    // host-API pinned staging (pageable 2-D copies cost milliseconds)
    uint8_t *h_stage_in; size_t h_stage_in_cap;
    uint8_t *h_out; size_t h_out_cap;
    int *h_flag;                           // pinned copy of the kernel error flag
    // host-API output staging
    void *d_out_kps, *d_out_desc, *d_out_n; int out_cap; int out_batch;
    float *d_out_ur, *d_out_depth;
    // stereo scratch
    int *d_st_dist; size_t st_cap;         // SAD per left keypoint (or -1)
    void *d_st_entries; size_t st_ent_cap;  // row table entries of k_stereo_prep (one uint4 per right keypoint, see orbx_stereo.hip)
    void *scratch[8]; size_t scratch_cap[8]; // host-API upload buffers
    // pipelined stereo frames: copies ride their own streams so that frame i+1 uploads and frame i-1 downloads while frame i computes
    PipeSlot pipe[ORBX_PIPE_DEPTH]; hipStream_t copy_in, copy_out; unsigned pipe_next;
    int pipe_warm_w, pipe_warm_h;    // the image size every lane / slot of the pipelined forms has been run on once (orbx_pipeline_warm); 0 = none yet
    bool pipe_counted;               // this handle is counted in the process-wide number of pipelined handles
    orbx_extractor *lanes[ORBX_PIPE_DEPTH - 1];   // further kernel lanes of the pipelined forms (shadow handles with their own stream and
                                     // workspaces): submission i runs on lane i % pipe_lanes, so the launch chains of neighbouring frames
                                     // overlap (a frame alone fills a few percent of the chip)
    int pipe_lanes;                  // lanes in use (1 .. ORBX_PIPE_DEPTH; ORBX_PIPE_LANES)
    bool pipe_inline;                // a frame's upload and download ride its lane's stream (no copy streams, no events between them)
    bool pipe_kcopy;                 // ... and are done by a copy kernel reading / writing the pinned buffers over PCIe (inline form only)
    int *flag_out;                   // where the next extraction's k_desc drops the kernel error flag (pipelined frames: into the slot's block)
    // grouped pyramid of small launches (k_pyr_group): group i builds levels first .. first + n - 1 in one launch
    struct PyrGroup { int first, n, tiles_x, tiles_y, tab_cx, tab_cy, lds_b, lds_bytes; } pyr_groups[ORBX_MAX_LEVELS];
    int n_pyr_groups;                // 0: this geometry has no grouped form (per-level launches at every batch size)
    int pyr_group_max_images;        // launches of at most this many images take the grouped form
    int pyr_group_mid_images;        // ... and up to this many: levels of the first group by one launch each, the later groups grouped
    int pyr_group_mid_cfg;           // the configured value of pyr_group_mid_images (default 24 / ORBX_PYR_GROUP_MID_IMAGES): what a non-zero group limit restores
    int stereo_kpw_forced;           // ORBX_STEREO_KPW (1 / 4), read once when the handle is created: forces k_stereo's keypoints per wave; 0 = by launch size
    int fast_waves;                  // waves per FAST cell: 0 = by launch size (several for a frame or two, one for batches); 1-4 forces (ORBX_FAST_WAVES)
    // stereo row table written by the most recent extraction as a by-product of k_desc (desc_rowtab): valid for the keypoint buffer
    // rt_kps (capacity rt_cap per image, rt_batch images); d_rt_off == nullptr: this geometry has none (more rows than k_desc's LDS holds)
    int *d_rt_off; size_t rt_off_cap; uint8_t *d_rt_entries; size_t rt_entries_cap; int rt_ent_cap;
    const void *rt_kps; int rt_cap, rt_batch;
    int *d_st_arrive;                // per stereo pair: workgroups of k_stereo that have finished (the last one applies the median cut)
    // state of the most recent extract
    const uint8_t *last_img0; size_t last_img_stride, last_pitch; int last_batch;
    // profiling
    bool prof, prof_chain; unsigned prof_mask; hipStream_t prof_last_stream; std::vector<ProfEvent> prof_ev; std::vector<hipEvent_t> prof_pool;
    float prof_ms[ORBX_STAGE_COUNT]; int prof_n[ORBX_STAGE_COUNT];
};

// one side of a BoW search on the device (FeatureVector as CSR, see orbx_featset)
struct DevFeat {
    int n, nnodes;
    const uint32_t *desc;     // [n][8]
    const uint32_t *node_id;  // [nnodes]
    const int32_t *node_off;  // [nnodes+1]
    const uint32_t *feat;
    const uint8_t *flag;
    const float *angle, *x, *y, *u_right;
    const int32_t *octave;
    const uint32_t *sdesc;    // [len(feat)][8] descriptors in list order (desc[feat[k]])
    const uint8_t *sflag;     // [len(feat)] flags in list order
};

// device-resident BoW data of a batch of frames (orbx_bow_transform_batch_device -> orbx_bowdb_search_batch_device)
struct orbx_bow_frames {
    int device, batch, cap, npad;
    uint8_t *d_buf;
    uint32_t *word_id; double *word_w; uint32_t *node_id;      // [batch][cap]
    uint32_t *bow_id; double *bow_val; int *counts;             // [batch][cap], [batch][cap], [batch][2] (nbow, nnodes)
    uint32_t *fv_node_id; int32_t *fv_node_off; uint32_t *fv_feat; // [batch][cap], [batch][cap+1 -> stride cap+4], [batch][npad]
    float *angle;                                               // [batch][cap]
    uint32_t *sdesc; uint8_t *sflag;                            // [batch][npad][8], [batch][npad]: list-order copies
    DevFeat *d_feats;                                           // [batch]
    uint8_t *h_buf; size_t h_cap;                               // pinned staging of orbx_bow_frames_read
    hipStream_t last_stream;                                    // stream of the most recent transform: the default of read / search
};

void orbx_set_error(const char *fmt, ...);
// orbx_bow.hip: CSR / pointer validation of a feature set; the legacy SearchByBoW kernels behind orbx_match.hip's entry points;
// the form forced by orbx_debug_set_bow_form (0 = none)
int orbx_feat_validate(const orbx_featset *f, int need_geom, int need_flag);
int orbx_bow_run_legacy(int mode, int device, const orbx_featset *as, int na, const orbx_featset *b,
                        float nnratio, int check_ori, int32_t *match, int *nmatches);
int orbx_bow_forced_form();
#define ORBX_HIP(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { \
    orbx_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); return ORBX_E_HIP; } } while (0)

// hipSetDevice only when the calling thread is on another device (a read of the thread's current device is cheaper than the set, and the
// hot host paths -- a pipelined submit makes three API entries per frame -- call it every time)
static inline hipError_t orbx_use_device(int device)
{
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess && cur == device) return hipSuccess;
    return hipSetDevice(device);
}

// host-side exact arithmetic helpers shared by the .hip files
static inline int orbx_cv_round(float v) { return (int)lrintf(v); }

int orbx_prepare_geometry(orbx_extractor *e, int w, int h);
void orbx_prof_begin(orbx_extractor *e, int stage, hipStream_t s);
void orbx_prof_end(orbx_extractor *e, hipStream_t s);
int orbx_ensure_out_staging(orbx_extractor *e, int batch, int cap);
int orbx_scratch(orbx_extractor *e, int slot, size_t bytes, void **out);
// waits for everything launched on this handle's workspaces (its own stream and the caller stream of the most recent
// launches): called before any geometry rebuild or workspace reallocation, so tables are never rewritten and buffers
// never freed under running kernels
int orbx_quiesce(orbx_extractor *e);
void orbx_pipe_handle_released();
// the handle's workspaces are about to be used by launches on stream s: if the previous launches went to another stream, s first
// waits for them (an event, no host synchronisation); records s as the handle's stream
int orbx_use_stream(orbx_extractor *e, hipStream_t s);

// level-0 / level-l pixel pointer of image b (device side helper)
struct PyrRef {
    const uint8_t *img0; long long img0_stride; int img0_pitch;
    const uint8_t *pyr; long long pyr_stride;
};
__device__ __forceinline__ const uint8_t *orbx_level_ptr(const PyrRef &p, const LevelGeom &L, int level, int b, int *pitch)
{
    if (level == 0) { *pitch = p.img0_pitch; return p.img0 + (long long)b * p.img0_stride; }
    *pitch = L.pitch;
    return p.pyr + (long long)b * p.pyr_stride + L.pyr_off;
}
