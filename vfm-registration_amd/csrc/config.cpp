// config.cpp -- vfm_config_*: caller-owned kernel policy bound per thread (csrc/config.h, include/vfmreg.h)
#include "common.h"
#include "config.h"

#include <string.h>

#include <new>

struct vfm_config {
    VfmConfig c;
};

static const VfmConfig k_factory{};
static thread_local const VfmConfig* t_bound = nullptr;

const VfmConfig& vfm_cfg() { return t_bound ? *t_bound : k_factory; }

namespace {
struct Field {
    const char* name;
    int VfmConfig::*p;
};
const Field k_fields[] = {
    {"coarse_slices", &VfmConfig::force_slices},     {"match_stats", &VfmConfig::match_stats},
    {"i8_min_queries", &VfmConfig::i8_min_queries},  {"prep_grid", &VfmConfig::prep_grid},
    {"ransac_exact_only", &VfmConfig::ransac_exact_only}, {"ransac_fused", &VfmConfig::ransac_fused},
    // the fields behind the compound keys, readable (and writable) one by one
    {"coarse_qsets", &VfmConfig::coarse_qsets},      {"seed_units", &VfmConfig::seed_units},
    {"select_variant", &VfmConfig::select_variant},  {"mx6_t4", &VfmConfig::mx6_t4},
    {"mx6_tune", &VfmConfig::mx6_tune},
    {"mx6_ns3", &VfmConfig::mx6_ns3},                {"prep_form", &VfmConfig::prep_stream},
    {"finish_short", &VfmConfig::finish_short},      {"rescan_rows", &VfmConfig::rescan_rows},
    {"vit_preprocess_patch", &VfmConfig::vit_preprocess_patch}, {"vit_xcd", &VfmConfig::vit_xcd},
    {"vit_cfg_narrow", &VfmConfig::vit_cfg_narrow},  {"vit_cfg_wide", &VfmConfig::vit_cfg_wide},
    {"vit_wpw", &VfmConfig::vit_wpw},                {"vit_hot_a", &VfmConfig::vit_hot_a},
    {"vit_wide_tile", &VfmConfig::vit_wide_tile},    {"vit_lds_shape", &VfmConfig::vit_lds_shape},
    {"vit_att_lds_min", &VfmConfig::vit_att_lds_min}, {"vit_lds_min_wg", &VfmConfig::vit_lds_min_wg},
    {"vit_astat_min", &VfmConfig::vit_astat_min},    {"vit_astat_two", &VfmConfig::vit_astat_two},
    {"vit_astat_nw", &VfmConfig::vit_astat_nw},      {"vit_fused_qkv", &VfmConfig::vit_fused_qkv},
    {"vit_trace_fused", &VfmConfig::vit_trace_fused},  {"vit_fused_mlp", &VfmConfig::vit_fused_mlp},
    {"voxel_replay2", &VfmConfig::voxel_replay2},    {"voxel_one_launch", &VfmConfig::voxel_small},
    {"voxel_trace", &VfmConfig::voxel_trace},        {"voxel_grid_ppt", &VfmConfig::voxel_grid_ppt},
};

// the code tables of round 1 - 5's vfm_debug_set_coarse_variant / _vit_gemm / _voxel_small, on a config
void set_coarse_variant(VfmConfig& c, int v) {
    if (v == 60 || v == 61) { c.rescan_rows = v == 61 ? 1 : 0; return; }
    if (v == 50 || v == 51) { c.finish_short = v == 51 ? 1 : 0; return; }
    if (v >= 40 && v <= 43) { c.prep_stream = v == 43 ? 3 : v == 42 ? 2 : v == 41 ? 1 : 0; return; }
    if (v == 32 || v == 33) { c.mx6_ns3 = v == 33 ? 1 : 0; return; }
    if (v == 30 || v == 31) { c.mx6_t4 = v == 30 ? 1 : 0; return; }
    c.seed_units = v == 7 ? 0 : 1;
    if (v == 7) v = 0;
    c.select_variant = v == 20 ? 1 : (v == 21 ? 2 : 0);
    if (v == 20 || v == 21) v = 0;
    c.coarse_qsets = v;
}
void set_vit_gemm(VfmConfig& c, int narrow, int wide) {
    switch (narrow) {
        case -3: case -4: c.vit_xcd = narrow == -3 ? 1 : 0; return;
        case -5: c.vit_lds_min_wg = wide; return;
        case -6: c.vit_lds_shape = wide; return;
        case -7: c.vit_att_lds_min = wide; return;
        case -8: c.vit_wpw = wide; return;
        case -9: c.vit_astat_min = wide; return;
        case -10: c.vit_astat_nw = wide; return;
        case -11: case -12: {
            unsigned long long v = (unsigned long long)(uintptr_t)c.vit_astat_dbg;
            v = narrow == -11 ? ((v & 0xffffffff00000000ull) | (unsigned)wide) : ((v & 0xffffffffull) | ((unsigned long long)(unsigned)wide << 32));
            c.vit_astat_dbg = reinterpret_cast<unsigned long long*>((uintptr_t)v);
            return;
        }
        case -14: c.vit_preprocess_patch = wide; return;
        case -15: c.vit_astat_two = wide; return;
        case -16: c.vit_hot_a = wide; return;
        case -17: c.vit_wide_tile = wide; return;
        default:
            c.vit_cfg_narrow = narrow ? narrow : 108;
            c.vit_cfg_wide = wide ? wide : 108;
    }
}
void set_voxel_small(VfmConfig& c, int on) {
    if (on == 2 || on == 3) c.voxel_replay2 = on == 3;
    else if (on >= 10 && on <= 10 + 64) c.voxel_grid_ppt = on - 10;
    else if (on == 100 || on == 101) c.voxel_trace = on - 100;
    else c.voxel_small = on;
}
}  // namespace

VFM_EXPORT int vfm_config_create(vfm_config_t** out) {
    VFM_CHECK_ARG(out, "config_create: null pointer");
    *out = new (std::nothrow) vfm_config();
    if (!*out) return vfm_fail(VFM_EINVAL, "config_create: out of memory");
    return VFM_OK;
}
VFM_EXPORT int vfm_config_destroy(vfm_config_t* cfg) {
    if (cfg && t_bound == &cfg->c) t_bound = nullptr;   // (other threads that still have it bound are the caller's to unbind first)
    delete cfg;
    return VFM_OK;
}
VFM_EXPORT int vfm_config_use(const vfm_config_t* cfg) {
    t_bound = cfg ? &cfg->c : nullptr;
    return VFM_OK;
}
VFM_EXPORT int vfm_config_set(vfm_config_t* cfg, const char* key, int64_t value) {
    VFM_CHECK_ARG(cfg && key, "config_set: null pointer");
    if (!strcmp(key, "coarse_variant")) { set_coarse_variant(cfg->c, (int)value); return VFM_OK; }
    if (!strcmp(key, "voxel_small")) { set_voxel_small(cfg->c, (int)value); return VFM_OK; }
    if (!strcmp(key, "vit_gemm")) { set_vit_gemm(cfg->c, (int)(value >> 32), (int)(int32_t)(uint32_t)(value & 0xffffffffll)); return VFM_OK; }
    for (const Field& f : k_fields)
        if (!strcmp(key, f.name)) {
            cfg->c.*(f.p) = (int)value;
            return VFM_OK;
        }
    return vfm_fail(VFM_EINVAL, "config_set: unknown key '%s'", key);
}
VFM_EXPORT int vfm_config_get(const vfm_config_t* cfg, const char* key, int64_t* value) {
    VFM_CHECK_ARG(key && value, "config_get: null pointer");
    const VfmConfig& c = cfg ? cfg->c : vfm_cfg();   // NULL: what the calling thread's entry points would read now
    for (const Field& f : k_fields)
        if (!strcmp(key, f.name)) {
            *value = c.*(f.p);
            return VFM_OK;
        }
    return vfm_fail(VFM_EINVAL, "config_get: unknown key '%s'", key);
}
