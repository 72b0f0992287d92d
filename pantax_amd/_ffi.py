"""ctypes binding of include/pantax_hip.h (libpantax_hip.so).

This is the same stub a Rust `extern "C"` block would mirror.  There is no CPU
fallback: if the library or a gfx950 device is missing, loading/`init` raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libpantax_hip.so")
_lib = None

# every symbol include/pantax_hip.h declares
SYMBOLS = [
    "pantax_hip_init", "pantax_hip_destroy", "pantax_hip_last_error", "pantax_hip_version", "pantax_hip_set_option",
    "pantax_hip_db_upload", "pantax_hip_db_upload_parts", "pantax_hip_db_free", "pantax_hip_reads_upload", "pantax_hip_reads_free",
    "pantax_hip_bin_reads", "pantax_hip_species_profile", "pantax_hip_db_reset", "pantax_hip_abundance_filter",
    "pantax_hip_trio_index", "pantax_hip_trio_get", "pantax_hip_node_coverage",
    "pantax_hip_strain_profile", "pantax_hip_pao_solve", "pantax_hip_pao_solve_batch", "pantax_hip_profile", "pantax_hip_profile_step", "pantax_hip_profile_step_enqueue", "pantax_hip_profile_step_collect", "pantax_hip_trio_index_prefetch", "pantax_hip_sort_rows",
    "pantax_hip_sample_ranks", "pantax_hip_chacha_block", "pantax_hip_gaf_filter", "pantax_hip_db_save_images", "pantax_hip_db_load_images",
    "pantax_hip_gaf_load", "pantax_hip_gaf_load_device", "pantax_hip_reads_load_gaf", "pantax_hip_reads_set_flags", "pantax_hip_gaf_view", "pantax_hip_gaf_free",
    "pantax_hip_graph_load", "pantax_hip_graph_view", "pantax_hip_graph_free", "pantax_hip_format_f64",
    "pantax_hip_reads_route_pack", "pantax_hip_route_buffer", "pantax_hip_route_free", "pantax_hip_reads_from_routed",
    "pantax_hip_timing_enable", "pantax_hip_timing_filter", "pantax_hip_timing_reset", "pantax_hip_timing_get", "pantax_hip_sync",
]


class PantaxHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("pantax_hip error %d: %s" % (code, msg))
        self.code = code


class Graphs(C.Structure):
    _fields_ = [("n_species", C.c_uint32), ("range_start", C.c_void_p), ("range_end", C.c_void_p),
                ("node_off", C.c_void_p), ("node_len", C.c_void_p), ("hap_off", C.c_void_p),
                ("path_off", C.c_void_p), ("path_nodes", C.c_void_p)]


class GraphPart(C.Structure):
    _fields_ = [("n_nodes", C.c_uint64), ("n_haps", C.c_uint64), ("node_len", C.c_void_p), ("path_off", C.c_void_p), ("path_nodes", C.c_void_p)]


class PackedReads(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("n_steps", C.c_uint64), ("step_off", C.c_void_p), ("node_id", C.c_void_p),
                ("pstart", C.c_void_p), ("pend", C.c_void_p), ("qlen", C.c_void_p), ("mapq", C.c_void_p),
                ("flags", C.c_void_p)]


class HapMetrics(C.Structure):
    _fields_ = [("has", C.c_uint32), ("is_rescue", C.c_int32), ("unique_trio_nodes_fraction", C.c_double),
                ("frequencies_mean", C.c_double), ("path_cov_ratio", C.c_double), ("first_sol", C.c_double),
                ("divergence", C.c_double), ("second_sol", C.c_double), ("total_cov_diff", C.c_double)]


class StrainConfig(C.Structure):
    _fields_ = [("unique_trio_nodes_fraction", C.c_double), ("unique_trio_nodes_mean_count_f", C.c_double),
                ("single_cov_ratio", C.c_double), ("min_depth", C.c_int64), ("shift", C.c_int32),
                ("sample_nodes", C.c_int32), ("solver_semantics", C.c_int32)]


class StepConfig(C.Structure):
    _fields_ = [("unique_trio_nodes_fraction", C.c_double), ("unique_trio_nodes_mean_count_f", C.c_double),
                ("single_cov_ratio", C.c_double), ("single_cov_diff", C.c_double), ("min_cov", C.c_int64),
                ("min_depth", C.c_int64), ("shift", C.c_int32), ("filtered", C.c_int32), ("sample_nodes", C.c_int32),
                ("rebuild_trio", C.c_int32), ("solver_semantics", C.c_int32)]


class SolveInfo(C.Structure):
    _fields_ = [("n_candidates", C.c_int32), ("status1", C.c_int32), ("status2", C.c_int32), ("iters1", C.c_int32),
                ("iters2", C.c_int32), ("n_rows", C.c_uint32), ("n_patterns", C.c_uint32), ("obj1", C.c_double),
                ("obj2", C.c_double)]


class SpeciesBatch(C.Structure):
    _fields_ = [("n_species", C.c_uint32), ("node_off", C.c_void_p), ("node_len", C.c_void_p), ("node_abundance", C.c_void_p),
                ("node_base_cov", C.c_void_p), ("hap_off", C.c_void_p), ("path_off", C.c_void_p), ("path_nodes", C.c_void_p),
                ("cand_off", C.c_void_p), ("cand_path_idx", C.c_void_p), ("fixed_zero", C.c_void_p)]


class SolutionBatch(C.Structure):
    _fields_ = [("x", C.c_void_p), ("path_cov_ratio", C.c_void_p), ("obj", C.c_void_p), ("status", C.c_void_p), ("iters", C.c_void_p)]


class ProfilingConfig(C.Structure):
    _fields_ = [("db", C.c_char_p), ("wd", C.c_char_p), ("output_dir", C.c_char_p), ("genomes_metadata", C.c_char_p),
                ("range_file", C.c_char_p), ("input_aln_file", C.c_char_p), ("species_len_file", C.c_char_p),
                ("out_binning_file", C.c_char_p), ("reads_binning_file", C.c_char_p),
                ("min_species_abundance", C.c_double), ("unique_trio_nodes_fraction", C.c_double),
                ("unique_trio_nodes_mean_count_f", C.c_double), ("single_cov_ratio", C.c_double),
                ("single_cov_diff", C.c_double), ("min_cov", C.c_int64), ("min_depth", C.c_int64),
                ("species", C.c_int32), ("strain", C.c_int32), ("shift", C.c_int32), ("filtered", C.c_int32),
                ("full", C.c_int32), ("force", C.c_int32), ("mode", C.c_int32), ("sample_nodes", C.c_int32),
                ("designated_species", C.c_char_p), ("zip", C.c_char_p), ("rank", C.c_int32), ("world_size", C.c_int32), ("image_cache", C.c_int32),
                ("allreduce_sum", C.c_void_p), ("comm_user", C.c_void_p), ("alltoallv", C.c_void_p), ("comm_device_buffers", C.c_int32),
                ("sample_test", C.c_int32), ("solver_semantics", C.c_int32), ("minimization_min_cov", C.c_double)]


# int (*allreduce_sum)(void *user, double *buf, uint64_t n)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_uint64)
# int (*alltoallv)(void *user, const void *send, const uint64_t *send_off, void *recv, const uint64_t *recv_off)
ALLTOALLV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p, C.POINTER(C.c_uint64))


def load():
    """dlopen libpantax_hip.so (built by __graft_entry__.build() / make -C pantax_amd/csrc)."""
    global _lib
    if _lib is None:
        # PANTAX_HIP_LIB: another build of the SAME library (A/B measurements, a -DLAD_PROFILE build) without touching the product file
        path = os.environ.get("PANTAX_HIP_LIB") or LIB_PATH
        if not os.path.exists(path):
            raise ImportError("%s is not built (run `python -c 'import __graft_entry__ as g; g.build()'`); "
                              "pantax_amd has no CPU fallback" % path)
        _lib = C.CDLL(path)
        _lib.pantax_hip_last_error.restype = C.c_char_p
        _lib.pantax_hip_last_error.argtypes = [C.c_void_p]
        _lib.pantax_hip_version.restype = C.c_char_p
        _lib.pantax_hip_destroy.restype = None
        _lib.pantax_hip_db_free.restype = None
        _lib.pantax_hip_reads_free.restype = None
        _lib.pantax_hip_gaf_free.restype = None
        _lib.pantax_hip_graph_free.restype = None
        _lib.pantax_hip_route_free.restype = None
    return _lib


def p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def as_c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)
