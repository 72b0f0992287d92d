// api_io.cpp -- C ABI over the host readers (a1 GAF tokenizer, a6 graph loaders). Host only.
#include <cstring>
#include <exception>
#include <memory>
#include <string>
#include <vector>
#include "../../include/pantax_hip.h"
#include "host_io.hpp"

using namespace ptx;

struct pantax_hip_graph { HostGraph g; std::vector<const char *> names; };

static thread_local std::string g_io_err;
static int io_fail(const char **err_out, const std::string &msg) {
    g_io_err = msg;
    if (err_out) *err_out = g_io_err.c_str();
    return PANTAX_HIP_E_IO;
}

extern "C" {

int pantax_hip_gaf_load(const char *path, int n_threads, pantax_hip_gaf **out, const char **err_out) {
    if (!path || !out) return PANTAX_HIP_E_INVALID;
    *out = nullptr;
    try {   // nothing throws across the boundary (allocation failures of a large file included)
        std::unique_ptr<pantax_hip_gaf> g(new pantax_hip_gaf());
        std::string e = g->mf.open(path);
        if (e.empty()) e = parse_gaf(g->mf, g->reads, n_threads);
        if (!e.empty()) return io_fail(err_out, e);
        *out = g.release();
        return 0;
    } catch (const std::exception &ex) {
        return io_fail(err_out, std::string("gaf_load: ") + ex.what());
    }
}
int pantax_hip_gaf_view(const pantax_hip_gaf *gaf, pantax_hip_packed_reads *v) {
    if (!gaf || !v) return PANTAX_HIP_E_INVALID;
    const HostReads &r = gaf->reads;
    const bool walks = r.pstart.size() == r.qlen.size();   // pantax_hip_reads_load_gaf keeps the walks on the device
    v->n_reads = r.qlen.size(); v->n_steps = r.node_id.size();
    v->step_off = walks ? r.step_off.data() : nullptr; v->node_id = walks ? r.node_id.data() : nullptr;
    v->pstart = walks ? r.pstart.data() : nullptr; v->pend = walks ? r.pend.data() : nullptr;
    v->qlen = r.qlen.data(); v->mapq = r.mapq.data(); v->flags = r.flags.data();
    return 0;
}
void pantax_hip_gaf_free(pantax_hip_gaf *gaf) { delete gaf; }

int pantax_hip_graph_load(const char *path, int format, pantax_hip_graph **out, const char **err_out) {
    if (!path || !out) return PANTAX_HIP_E_INVALID;
    *out = nullptr;
    try {
        std::unique_ptr<pantax_hip_graph> g(new pantax_hip_graph());
        std::string e = format == 1 ? read_graph_bin(path, g->g) : (format == 2 || format == 3) ? read_graph_zip(path, format, g->g) : read_gfa(path, g->g);
        if (!e.empty()) return io_fail(err_out, e);
        for (auto &n : g->g.hap_names) g->names.push_back(n.c_str());
        *out = g.release();
        return 0;
    } catch (const std::exception &ex) {
        return io_fail(err_out, std::string("graph_load: ") + ex.what());
    }
}
int pantax_hip_graph_view(const pantax_hip_graph *g, uint64_t *n_nodes, uint64_t *n_haps, const int64_t **node_len,
                          const uint64_t **path_off, const uint32_t **path_nodes, const char *const **hap_names) {
    if (!g) return PANTAX_HIP_E_INVALID;
    if (n_nodes) *n_nodes = g->g.node_len.size();
    if (n_haps) *n_haps = g->g.hap_names.size();
    if (node_len) *node_len = g->g.node_len.data();
    if (path_off) *path_off = g->g.path_off.data();
    if (path_nodes) *path_nodes = g->g.path_nodes.data();
    if (hap_names) *hap_names = g->names.data();
    return 0;
}
void pantax_hip_graph_free(pantax_hip_graph *g) { delete g; }

int pantax_hip_format_f64(double v, char *buf, size_t cap) {
    if (!buf || cap == 0) return PANTAX_HIP_E_INVALID;
    const std::string s = fmt_f64(v);
    if (s.size() + 1 > cap) return PANTAX_HIP_E_LIMIT;
    std::memcpy(buf, s.c_str(), s.size() + 1);
    return (int)s.size();
}

}  // extern "C"
