// Layout of the concatenated weight vector (order of packppi_amd/weights.py::weight_spec()).  Host-only C++, no HIP headers:
// shared by the library (pp_internal.h) and by the CPU-built sanitizer harness (tests/native/host_sanitize.cpp).
#pragma once
#include <stddef.h>

// the five power-of-two operand-scale vectors behind small LayerNorm gains (pp_rebalance.h ln_operand_scales, plan->ln_scale)
#define PP_LN_E0 0        /* h_E0 -> layer 0's two W_B blocks (k_edge_static) */
#define PP_LN_E(l) (1 + (l))      /* h_E after layer l -> layer l + 1's W_B blocks */
#define PP_LN_X1(l) (3 + (l))     /* x1 of layer l -> its edge FFN's W_in */

// ---------------------------------------------------------------------------------------------
// Offsets (in floats) into the concatenated weight buffer, order of weights.py::weight_spec().
// ---------------------------------------------------------------------------------------------
struct LayerOff {
    size_t pts_node_w, pts_node_b, pts_edge_w, pts_edge_b;
    size_t nm_in_w, nm_in_b, nm_mid_w, nm_mid_b, nm_out_w, nm_out_b;   // node_message_fn
    size_t em_in_w, em_in_b, em_mid_w, em_mid_b, em_out_w, em_out_b;   // edge_message_fn
    size_t norm_g[4], norm_b[4];
    size_t nd_in_w, nd_in_b, nd_out_w, nd_out_b;                       // node_dense
    size_t ed_in_w, ed_in_b, ed_out_w, ed_out_b;                       // edge_dense
};
struct WeightOff {
    size_t node_emb_w, node_emb_b, norm_nodes_g, norm_nodes_b;
    size_t edge_emb_w, edge_emb_b, norm_edges_g, norm_edges_b;
    LayerOff layer[3];
    size_t d0_in_w, d0_in_b, d0_out_w, d0_out_b, d2_in_w, d2_in_b, d2_out_w, d2_out_b;
    size_t total;
};


inline WeightOff pp_weight_offsets() {
    WeightOff o;
    size_t p = 0;
    auto take = [&](size_t n) { size_t r = p; p += n; return r; };
    o.node_emb_w = take(128 * 51); o.node_emb_b = take(128);
    o.norm_nodes_g = take(128); o.norm_nodes_b = take(128);
    o.edge_emb_w = take(128 * 468); o.edge_emb_b = take(128);
    o.norm_edges_g = take(128); o.norm_edges_b = take(128);
    for (int l = 0; l < 3; l++) {
        LayerOff &L = o.layer[l];
        L.pts_node_w = take(24 * 128); L.pts_node_b = take(24);
        L.pts_edge_w = take(24 * 128); L.pts_edge_b = take(24);
        L.nm_in_w = take(128 * 456); L.nm_in_b = take(128);
        L.nm_mid_w = take(128 * 128); L.nm_mid_b = take(128);
        L.nm_out_w = take(128 * 128); L.nm_out_b = take(128);
        L.em_in_w = take(128 * 456); L.em_in_b = take(128);
        L.em_mid_w = take(128 * 128); L.em_mid_b = take(128);
        L.em_out_w = take(128 * 128); L.em_out_b = take(128);
        for (int k = 0; k < 4; k++) { L.norm_g[k] = take(128); L.norm_b[k] = take(128); }
        L.nd_in_w = take(512 * 128); L.nd_in_b = take(512);
        L.nd_out_w = take(128 * 512); L.nd_out_b = take(128);
        L.ed_in_w = take(512 * 128); L.ed_in_b = take(512);
        L.ed_out_w = take(128 * 512); L.ed_out_b = take(128);
    }
    o.d0_in_w = take(64 * 128); o.d0_in_b = take(64);
    o.d0_out_w = take(32 * 64); o.d0_out_b = take(32);
    o.d2_in_w = take(16 * 32); o.d2_in_b = take(16);
    o.d2_out_w = take(4 * 16); o.d2_out_b = take(4);
    o.total = p;
    return o;
}
