// config.h -- kernel policy of libvfmreg_hip.so as a CALLER-OWNED object (round 6; VERDICT r5 item 7, SURVEY.md 8 B.5: "no global state
// except the last-error string").  Until round 5 the A/B switches of the test and bench tooling were process-global ints written by
// vfm_debug_set_*: two pipelines of one process with different settings raced.  Now: a vfm_config_t holds them (factory settings at
// creation), vfm_config_use() binds one to the CALLING THREAD -- thread-local, like the last-error string -- and every entry point reads
// its policy from the calling thread's binding at call time.  No binding = the immutable factory settings.
#pragma once

struct VfmConfig {
    // ---- matcher (match_api.hip, match_prep.hip, match_coarse_*.hip, match_finish.hip)
    int force_slices = 0;      // map slices of the coarse pass (0 = the launchers' rules)
    int coarse_qsets = 0;      // "coarse_variant" 0 / 1 / 2 / 4 / 5 / 10 / 12: see include/vfmreg.h, vfm_config_set
    int seed_units = 1;        // "coarse_variant" 7: no seed units
    int select_variant = 0;    // "coarse_variant" 20 / 21: general select kernel / no chunk-major rescan
    int mx6_t4 = 1;            // "coarse_variant" 30 / 31: fused fp6 half-width kernel with one (default) / two chunks per barrier
    int mx6_ns3 = 1;           // "coarse_variant" 32 / 33: ... with two / three (default) query tiles per wave at d = 384
    int prep_stream = 3;       // "coarse_variant" 40 .. 43: fp6 operand preparation by prep_chunk_kernel (0) / prep_stream_kernel (1) / by width (2) / prep_once_kernel (3, default)
    int finish_short = 0;      // "coarse_variant" 50 / 51: chunk-major rescan as long-lived (default) / short workgroups
    int rescan_rows = 1;       // "coarse_variant" 60 / 61: rescan gathers its queries from the fragment tiles / the row-major int8 scan (default)
    int mx6_tune = 0;          // (A/B) bit 0: s_setprio 1 for waves 4 - 7 of the fp6 coarse kernel; bit 1: its ring five steps deep (headline shape)
    int match_stats = 0;       // per-query counters of a search (they cost same-address atomics)
    int i8_min_queries = 0;    // the gated family takes the int8 pass for more than this many query rows
    int prep_grid = -1;        // workgroups of prep_chunk_kernel (-1 = one per 128-row group, 0 = one per compute unit, n > 0)
    // ---- RANSAC (ransac.hip)
    int ransac_exact_only = 0;   // 1 = every hypothesis scored in fp64 (no bounds)
    int ransac_fused = 2;        // 2 (default since round 6) = 8 launches, 1 = the 5-launch chain (slower: its one-workgroup gather), 0 = round 5's 11
    // ---- ViT (vit.hip): "vit_gemm" (narrow, wide) codes, see include/vfmreg.h
    int vit_preprocess_patch = 1, vit_xcd = 1, vit_cfg_narrow = 108, vit_cfg_wide = 108, vit_wpw = 0, vit_hot_a = 0, vit_wide_tile = 0,
        vit_lds_shape = 23, vit_att_lds_min = 1, vit_lds_min_wg = 256, vit_astat_min = 0, vit_astat_two = 1, vit_astat_nw = 0,
        vit_trace_fused = 0,   // (tools) whose trace buffer it is: 0 the GEMM kernels', 1 vit_qkv_attention_kernel's, 2 the proj launches' of the LDS-tiled kernel
        vit_fused_mlp = 0,     // fc1 -> GELU -> fc2 of 128 tokens in one workgroup: n > 0 from n images on, -1 never, 0 vfm_vit_forward's policy
        vit_fused_qkv = 0;   // QKV + attention of an (image, head) in one workgroup: n > 0 from n images on, -1 never, 0 vfm_vit_forward's policy
    unsigned long long* vit_astat_dbg = nullptr;   // (tools) device buffer of the token-stationary kernel's placement trace
    // ---- voxel containers (voxel.hip): "voxel_small" codes
    int voxel_replay2 = 1, voxel_small = 1, voxel_trace = 0, voxel_grid_ppt = 0;
};

// the calling thread's policy (its bound vfm_config_t, or the factory settings)
const VfmConfig& vfm_cfg();
