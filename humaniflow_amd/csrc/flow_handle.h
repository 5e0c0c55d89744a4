// Flow handle + host-side pack description shared by flow_pack.hip and flow_kernels.hip.
#pragma once
#include <vector>
#include "flow_common.h"

namespace hmf {

struct FlowPack {
    int feat_dim, nb, in_dim;
    float radius, base_std;
    int o_wisgc, o_bisgc, o_wbeta;
    std::vector<float> arena;
    std::vector<PartTab> tab;
    std::vector<int> anc_off;
    Levels lev;
    bool generic = false;            // non-default variant (or force_generic): the MFMA operand blocks are not packed
    int T = 2, transform_type = 0, permute_type = 0;
    std::vector<GenPartTab> gtab;
};

int build_flow_pack(const hmf_flow_host_t* w, FlowPack& fp);

}  // namespace hmf

struct hmf_flow {
    hmf::FlowDev d;
    hmf::GenDev g;
    std::vector<hmf::PartTab> host_tab;
    const int* d_ids;   // device table 0..22
    const int* d_map = nullptr;   // arena slot -> flat parameter position + 1 (0 = padding); hmf_flow_create_updatable only
    size_t arena_floats = 0, flat_count = 0;
    int gen_layers_tr = 0, gen_layers_plu = 0;   // variants: layers (hidden + 1) of a transform / PLU hyper-network (records of the reverse sweeps)
    void* mem;
};
