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
};

int build_flow_pack(const hmf_flow_host_t* w, FlowPack& fp);

}  // namespace hmf

struct hmf_flow {
    hmf::FlowDev d;
    std::vector<hmf::PartTab> host_tab;
    const int* d_ids;   // device table 0..22
    void* mem;
};
