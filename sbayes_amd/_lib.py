"""ctypes binding of the C-ABI engine library (include/sbe_engine.h: the drop-in boundary; sbe_engine_steps.h: the one-call
step family, no caller in the reference; sbe_engine_diag.h: self-tests and measurement hooks -- one library exports all three).

The product has NO CPU fallback: if the HIP library is missing or no GPU is usable the
loader / sbe_create raise -- nothing silently routes around the device."""
from __future__ import annotations

import ctypes as ct
import os
from pathlib import Path

LIB_NAME = "libsbe_engine.so"
ABI_VERSION = 6                    # SBE_ABI_VERSION of include/sbe_engine.h
_LIB = None

c_engine_p = ct.c_void_p
u8p = ct.POINTER(ct.c_uint8)
i32p = ct.POINTER(ct.c_int32)
i64p = ct.POINTER(ct.c_int64)
f32p = ct.POINTER(ct.c_float)
f64p = ct.POINTER(ct.c_double)


class SbeInfo(ct.Structure):
    _fields_ = [
        ("abi_version", ct.c_int32), ("device", ct.c_int32),
        ("n_objects", ct.c_int32), ("n_features", ct.c_int32), ("n_states", ct.c_int32),
        ("n_components", ct.c_int32), ("n_slots", ct.c_int32), ("n_groups_total", ct.c_int32),
        ("n_na", ct.c_int64), ("hbm_bytes", ct.c_int64), ("compute_units", ct.c_int32),
        ("device_name", ct.c_char * 64),
    ]


# name -> (restype, argtypes); mirrors the three headers under include/ one to one
PROTOTYPES = {
    "sbe_abi_version": (ct.c_int, []),
    "sbe_device_count": (ct.c_int, [ct.POINTER(ct.c_int)]),
    "sbe_last_error": (ct.c_char_p, [c_engine_p]),
    "sbe_create": (ct.c_int, [ct.POINTER(c_engine_p), ct.c_int, ct.c_int, ct.c_int, ct.c_int, ct.c_int,
                              i32p, ct.c_int, ct.c_void_p]),
    "sbe_destroy": (ct.c_int, [c_engine_p]),
    "sbe_get_info": (ct.c_int, [c_engine_p, ct.POINTER(SbeInfo)]),
    "sbe_get_na": (ct.c_int, [c_engine_p, ct.c_void_p]),
    "sbe_set_option": (ct.c_int, [c_engine_p, ct.c_int, ct.c_int]),
    "sbe_sync": (ct.c_int, [c_engine_p]),
    "sbe_component_lh": (ct.c_int, [c_engine_p, ct.c_void_p, ct.c_int, ct.c_int, ct.c_void_p, ct.c_void_p,
                                    ct.c_int, ct.c_void_p, ct.c_int64, ct.c_int64, ct.c_double]),
    "sbe_likelihood_per_component_exact": (ct.c_int, [c_engine_p, ct.c_int, ct.c_void_p]),
    "sbe_set_groups": (ct.c_int, [c_engine_p, ct.c_int, ct.c_int, ct.c_void_p]),
    "sbe_set_group_ids": (ct.c_int, [c_engine_p, ct.c_int, ct.c_int, ct.c_void_p]),
    "sbe_get_group_ids": (ct.c_int, [c_engine_p, ct.c_int, ct.c_int, ct.c_void_p]),
    "sbe_get_weights": (ct.c_int, [c_engine_p, ct.c_int, ct.c_void_p]),
    "sbe_set_source": (ct.c_int, [c_engine_p, ct.c_int, ct.c_void_p]),
    "sbe_set_source_rows": (ct.c_int, [c_engine_p, ct.c_int, ct.c_void_p, ct.c_int, ct.c_void_p]),
    "sbe_get_source_rows": (ct.c_int, [c_engine_p, ct.c_int, ct.c_void_p, ct.c_int, ct.c_void_p]),
    "sbe_recount": (ct.c_int, [c_engine_p, ct.c_int, ct.c_int]),
    "sbe_update_counts": (ct.c_int, [c_engine_p, ct.c_int, ct.c_int, ct.c_void_p, ct.c_int, ct.c_void_p]),
    "sbe_accumulate_counts": (ct.c_int, [c_engine_p, ct.c_int, ct.c_void_p, ct.c_int, ct.c_int, ct.c_void_p]),
    "sbe_set_counts": (ct.c_int, [c_engine_p, ct.c_int, ct.c_int, ct.c_void_p]),
    "sbe_get_counts": (ct.c_int, [c_engine_p, ct.c_int, ct.c_int, ct.c_void_p]),
    "sbe_get_counts_all": (ct.c_int, [c_engine_p, ct.c_int, ct.c_void_p]),
    "sbe_set_concentration": (ct.c_int, [c_engine_p, ct.c_int, ct.c_void_p, ct.c_int]),
    "sbe_update_probs": (ct.c_int, [c_engine_p, ct.c_int, ct.c_int, ct.c_double, ct.c_double, ct.c_void_p]),
    "sbe_update_probs_mask": (ct.c_int, [c_engine_p, ct.c_int, ct.c_uint, ct.c_double, ct.c_double, ct.c_void_p]),
    "sbe_set_probs": (ct.c_int, [c_engine_p, ct.c_int, ct.c_int, ct.c_void_p]),
    "sbe_get_probs": (ct.c_int, [c_engine_p, ct.c_int, ct.c_int, ct.c_void_p]),
    "sbe_set_weights": (ct.c_int, [c_engine_p, ct.c_int, ct.c_void_p]),
    "sbe_get_weights_normalized": (ct.c_int, [c_engine_p, ct.c_int, ct.c_void_p]),
    "sbe_likelihood_per_component": (ct.c_int, [c_engine_p, ct.c_int, ct.c_void_p]),
    "sbe_observation_lh": (ct.c_int, [c_engine_p, ct.c_int, ct.c_void_p]),
    "sbe_mixture_loglik": (ct.c_int, [c_engine_p, ct.c_int, f64p]),
    "sbe_mixture_loglik_batch": (ct.c_int, [c_engine_p, ct.c_int, ct.c_int, ct.c_void_p]),
    "sbe_mixture_loglik_batch_async": (ct.c_int, [c_engine_p, ct.c_int, ct.c_int]),
    "sbe_fetch_results": (ct.c_int, [c_engine_p, ct.c_int, ct.c_int, ct.c_void_p]),
    "sbe_collapsed_loglik": (ct.c_int, [c_engine_p, ct.c_int, ct.c_int, ct.c_void_p, ct.c_void_p]),
    "sbe_collapsed_loglik_all": (ct.c_int, [c_engine_p, ct.c_int, ct.c_void_p]),
    "sbe_normalize_tables": (ct.c_int, [c_engine_p, ct.c_void_p, ct.c_int, ct.c_void_p, ct.c_int, ct.c_double,
                                        ct.c_double, ct.c_void_p, ct.c_void_p]),
    "sbe_dirichlet_logpdf": (ct.c_int, [c_engine_p, ct.c_void_p, ct.c_int, ct.c_void_p, ct.c_int, ct.c_void_p,
                                        ct.c_void_p]),
    "sbe_effect_counts": (ct.c_int, [c_engine_p, ct.c_void_p, ct.c_int, ct.c_void_p, ct.c_void_p, ct.c_int,
                                     ct.c_void_p]),
    "sbe_normalize_weights": (ct.c_int, [c_engine_p, ct.c_void_p, ct.c_int, ct.c_void_p, ct.c_int, ct.c_void_p]),
    "sbe_cluster_marginals": (ct.c_int, [c_engine_p, ct.c_int, ct.c_void_p, ct.c_void_p, ct.c_int, ct.c_double,
                                         ct.c_void_p]),
    "sbe_jump_lh": (ct.c_int, [c_engine_p, ct.c_int, ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_int,
                               ct.c_double, ct.c_void_p]),
    "sbe_source_lh_by_feature": (ct.c_int, [c_engine_p, ct.c_int, ct.c_void_p]),
    "sbe_source_posterior": (ct.c_int, [c_engine_p, ct.c_int, ct.c_void_p, ct.c_int, ct.c_double, ct.c_double,
                                        ct.c_void_p]),
    "sbe_sample_source": (ct.c_int, [c_engine_p, ct.c_int, ct.c_int, ct.c_void_p, ct.c_int, ct.c_double, ct.c_double,
                                     ct.c_int, ct.c_void_p, ct.POINTER(ct.c_double), ct.c_void_p]),
    "sbe_source_logprob": (ct.c_int, [c_engine_p, ct.c_int, ct.c_int, ct.c_void_p, ct.c_int, ct.c_double, ct.c_double,
                                      ct.c_int, ct.POINTER(ct.c_double), ct.c_void_p]),
    "sbe_set_rng": (ct.c_int, [c_engine_p, ct.c_uint64, ct.c_uint64]),
    "sbe_get_rng": (ct.c_int, [c_engine_p, ct.POINTER(ct.c_uint64), ct.POINTER(ct.c_uint64)]),
    "sbe_test_philox": (ct.c_int, [c_engine_p, ct.c_void_p, ct.c_int, ct.c_void_p]),
    "sbe_subset_lh": (ct.c_int, [c_engine_p, ct.c_void_p, ct.c_int, ct.c_int, ct.c_void_p, ct.c_void_p, ct.c_int,
                                 ct.c_void_p, ct.c_double, ct.c_void_p]),
    "sbe_host_group_ids": (ct.c_int, [ct.c_void_p, ct.c_int, ct.c_int64, ct.c_void_p, ct.c_int, ct.c_int, ct.c_void_p]),
    "sbe_host_touched_groups": (ct.c_int, [ct.c_void_p, ct.c_void_p, ct.c_int64, ct.c_int, ct.c_void_p, ct.c_void_p]),
    "sbe_host_source_ids": (ct.c_int, [ct.c_void_p, ct.c_int64, ct.c_int, ct.c_int, ct.c_void_p, ct.c_int, ct.c_void_p]),
    "sbe_host_subset_ids": (ct.c_int, [ct.c_void_p, ct.c_int, ct.c_int64, ct.c_int, ct.c_int, ct.c_void_p, ct.c_void_p, ct.c_void_p,
                                       ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_void_p]),
    "sbe_host_diff_rows": (ct.c_int64, [ct.c_void_p, ct.c_void_p, ct.c_int64, ct.c_int64, ct.c_void_p]),
    "sbe_set_uniform_counts": (ct.c_int, [c_engine_p, ct.c_void_p]),
    "sbe_counts_delta": (ct.c_int, [c_engine_p, ct.c_void_p, ct.c_int, ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_void_p,
                                    ct.c_void_p, ct.c_int, ct.c_void_p]),
    "sbe_counts_delta_apply": (ct.c_int, [c_engine_p, ct.c_int, ct.c_int, ct.c_int, ct.c_void_p, ct.c_int, ct.c_void_p, ct.c_void_p, ct.c_void_p,
                                          ct.c_void_p, ct.c_void_p, ct.c_int, ct.c_void_p]),
    "sbe_set_counts_rows": (ct.c_int, [c_engine_p, ct.c_int, ct.c_void_p, ct.c_int, ct.c_void_p]),
    "sbe_set_slot_delta": (ct.c_int, [c_engine_p, ct.c_int, ct.c_int, ct.c_void_p, ct.c_void_p, ct.c_int, ct.c_void_p, ct.c_int,
                                      ct.c_void_p, ct.c_int, ct.c_void_p]),
    "sbe_set_counts_rows_probs": (ct.c_int, [c_engine_p, ct.c_int, ct.c_void_p, ct.c_int, ct.c_void_p]),
    "sbe_given_unchanged_lh": (ct.c_int, [c_engine_p, ct.c_int, ct.c_int, ct.c_void_p, ct.c_int, ct.c_double, ct.c_double,
                                          ct.c_void_p]),
    "sbe_given_unchanged_gibbs": (ct.c_int, [c_engine_p, ct.c_int, ct.c_int, ct.c_void_p, ct.c_int, ct.c_double, ct.c_double,
                                             ct.c_int, ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_void_p,
                                             ct.c_void_p]),
    "sbe_cluster_posterior_marginals": (ct.c_int, [c_engine_p, ct.c_int, ct.c_int, ct.c_double, ct.c_double, ct.c_void_p,
                                                   ct.c_int, ct.c_void_p]),
    "sbe_jump_lh_resident": (ct.c_int, [c_engine_p, ct.c_int, ct.c_int, ct.c_int, ct.c_double, ct.c_double, ct.c_void_p,
                                        ct.c_int, ct.c_void_p]),
    "sbe_source_prior": (ct.c_int, [c_engine_p, ct.c_int, ct.c_void_p]),
    "sbe_collapsed_and_source_prior": (ct.c_int, [c_engine_p, ct.c_int, ct.c_void_p, ct.c_void_p]),
    "sbe_observation_lh_exact": (ct.c_int, [c_engine_p, ct.c_int, ct.c_void_p]),
    "sbe_gibbs_step": (ct.c_int, [c_engine_p, ct.c_int, ct.c_int, ct.c_void_p, ct.c_int, ct.c_double, ct.c_double, ct.c_int,
                                  ct.c_void_p, ct.POINTER(ct.c_double), ct.POINTER(ct.c_double), ct.c_void_p,
                                  ct.POINTER(ct.c_double), ct.c_void_p]),
    "sbe_step_batch": (ct.c_int, [c_engine_p, ct.c_int, ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_void_p,
                                  ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_void_p]),
    "sbe_step_delta": (ct.c_int, [c_engine_p, ct.c_int, ct.c_int, ct.c_void_p, ct.c_void_p, ct.c_int, ct.c_void_p, ct.c_int, ct.c_void_p,
                                  ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_void_p]),
    "sbe_step_batch_delta": (ct.c_int, [c_engine_p, ct.c_int, ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_void_p,
                                        ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_void_p,
                                        ct.c_void_p]),
    "sbe_step": (ct.c_int, [c_engine_p, ct.c_int, ct.c_int, ct.c_void_p, ct.c_void_p, ct.c_int, ct.c_void_p, ct.c_void_p,
                            ct.c_void_p, ct.c_void_p, ct.c_void_p]),
    "sbe_test_fast_log": (ct.c_int, [c_engine_p, ct.c_void_p, ct.c_int, ct.c_void_p, ct.c_void_p]),
    "sbe_test_lgamma": (ct.c_int, [c_engine_p, ct.c_void_p, ct.c_int, ct.c_void_p]),
    "sbe_test_tab_log": (ct.c_int, [c_engine_p, ct.c_void_p, ct.c_int, ct.c_void_p]),
    "sbe_test_roundtrip": (ct.c_int, [c_engine_p, ct.c_int, ct.c_int]),
    "sbe_given_unchanged_gibbs_apply": (ct.c_int, [c_engine_p, ct.c_int, ct.c_int, ct.c_int, ct.c_void_p, ct.c_int, ct.c_double, ct.c_double, ct.c_int,
                                                   ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_void_p,
                                                   ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_void_p]),
    "sbe_given_unchanged_gibbs_counts": (ct.c_int, [c_engine_p, ct.c_int, ct.c_int, ct.c_void_p, ct.c_int, ct.c_double, ct.c_double, ct.c_int,
                                                    ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_void_p,
                                                    ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_void_p]),
    "sbe_gibbs_propose_supported": (ct.c_int, [c_engine_p]),
    "sbe_gibbs_propose": (ct.c_int, [c_engine_p, ct.c_int, ct.c_int, ct.c_void_p, ct.c_int, ct.c_double, ct.c_double, ct.c_int,
                                     ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_void_p]),
    "sbe_gibbs_propose_apply": (ct.c_int, [c_engine_p, ct.c_int, ct.c_int, ct.c_void_p, ct.c_int, ct.c_double, ct.c_double, ct.c_int,
                                           ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_void_p]),
    "sbe_copy_slot": (ct.c_int, [c_engine_p, ct.c_int, ct.c_int]),
    "sbe_timer_start": (ct.c_int, [c_engine_p]),
    "sbe_timer_stop": (ct.c_int, [c_engine_p, ct.POINTER(ct.c_float)]),
    "sbe_timer_mark": (ct.c_int, [c_engine_p]),
    "sbe_timer_elapsed": (ct.c_int, [c_engine_p, ct.POINTER(ct.c_float)]),
    "sbe_kernel_timing": (ct.c_int, [c_engine_p, ct.c_int, ct.POINTER(ct.c_int), ct.POINTER(ct.c_float)]),
    "sbe_last_mixture_kernel": (ct.c_char_p, [c_engine_p]),
    "sbe_profile_mixture": (ct.c_int, [c_engine_p, ct.c_int, ct.c_int, ct.c_int, ct.POINTER(ct.c_float),
                                       ct.POINTER(ct.c_float)]),
}


def lib_path() -> Path:
    env = os.environ.get("SBAYES_AMD_LIB")
    return Path(env) if env else Path(__file__).resolve().parent / LIB_NAME


def load():
    """Load the in-tree HIP library and attach prototypes.  Raises if it is missing."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not path.exists():
        raise RuntimeError(
            f"sbayes_amd: HIP engine library not found at {path}. Build it with "
            f"`python -c 'import __graft_entry__ as g; g.build()'` (or ./build.sh). "
            f"There is no CPU fallback.")
    lib = ct.CDLL(str(path))
    for name, (restype, argtypes) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError if the library lacks a declared symbol
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.sbe_abi_version() != ABI_VERSION:
        raise RuntimeError(f"sbayes_amd: ABI version mismatch ({lib.sbe_abi_version()} != {ABI_VERSION})")
    _LIB = lib
    return lib
