"""ctypes binding of libheal_amd.so -- the C ABI declared in include/heal_amd.h.

This is the only place the package touches the native library.  There is NO fallback: if the
library is missing or a call fails, an exception is raised.
"""
import ctypes
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HEAL_AMD_LIB") or os.path.join(HERE, "lib", "libheal_amd.so")  # env: A/B a rebuilt library
HEADER = os.path.join(os.path.dirname(HERE), "include", "heal_amd.h")
HEADER_EXPERIMENTAL = os.path.join(os.path.dirname(HERE), "include", "heal_amd_experimental.h")

_lib = None

c_void_p, c_int, c_float, c_size_t = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t

_SIGNATURES = {
    # name: (restype, argtypes)
    "heal_abi_version": (c_int, []),
    "heal_last_error": (ctypes.c_char_p, []),
    "heal_fill_bytes": (c_int, [c_void_p, c_int, c_size_t, c_void_p]),
    "heal_next_launch_events": (c_int, [c_void_p, c_void_p]),
    "heal_voxelize_workspace": (c_size_t, [c_int, c_int, c_int, ctypes.c_longlong]),
    "heal_voxelize": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int,
                              c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "heal_voxelize_batch_workspace": (c_size_t, [c_int, c_int, c_int, c_int, ctypes.c_longlong]),
    "heal_voxelize_batch": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "heal_mask_points": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "heal_pfn_scatter_workspace": (c_size_t, [c_int] * 5),
    "heal_pfn_scatter": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int,
                                 c_void_p, c_void_p, c_void_p, c_int,
                                 c_float, c_float, c_float, c_float, c_float, c_float,
                                 c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "heal_pfn_pillars": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                 c_float, c_float, c_float, c_float, c_float, c_float, c_int, c_int, c_int, c_void_p, c_void_p,
                                 c_void_p]),
    "heal_pillar_canvas": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "heal_pillar_stem_block": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int] + [c_void_p] * 4 + [c_int] + [c_void_p] * 3),
    "heal_bev_pool_backward": (c_int, [c_void_p] * 5 + [c_int] * 6 + [c_void_p] * 6),
    "heal_warp_fuse": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p,
                               c_void_p, c_void_p]),
    "heal_warp_fuse_levels": (c_int, [c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                      c_void_p, c_void_p, c_void_p]),
    "heal_warp_fuse_backward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p,
                                        c_void_p, c_void_p, c_void_p, c_void_p]),
    "heal_warp_agent": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p,
                                c_void_p, c_void_p, c_void_p]),
    "heal_warp_agents_pm": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "heal_fuse_warped": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "heal_fuse_warped_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "heal_decode_nms_workspace": (c_size_t, [c_int, c_int]),
    "heal_decode_nms": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                c_float, c_float, c_float, c_int, c_void_p, c_void_p,
                                c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "heal_quad_iou": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "heal_bev_pool_workspace": (c_size_t, [c_int] * 9),
    "heal_bev_pool": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                              c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "heal_bev_pool_pm_workspace": (c_size_t, [c_int] * 5),
    "heal_bev_pool_pm": (c_int, [c_void_p, c_int, c_void_p, c_void_p] + [c_int] * 6 + [c_void_p] * 5 + [c_size_t, c_void_p]),
    "heal_bev_pool_scatter": (c_int, [c_void_p, c_int, c_void_p, c_void_p] + [c_int] * 6 + [c_void_p] * 4 + [c_size_t, c_void_p]),
    "heal_bev_pool_scatter_multi": (c_int, [c_int] + [c_void_p] * 16),
    "heal_bev_pool_emit": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "heal_bev_stem_block": (c_int, [c_int, c_int] + [c_void_p] * 8 + [c_size_t, c_void_p]),
    "heal_pfn_train_blocks": (c_int, [c_int]),
    "heal_pfn_features": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_float, c_float,
                                  c_float, c_float, c_float, c_float, c_void_p, c_void_p]),
    "heal_pfn_moments": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_float, c_float, c_float, c_float, c_float,
                                 c_void_p, c_void_p]),
    "heal_pfn_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_float, c_float, c_float, c_float, c_float, c_float, c_void_p, c_void_p, c_void_p]),
    "heal_mean_vfe": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "heal_sp_sort_workspace": (c_size_t, [c_int]),
    "heal_sp_sort_sites": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p,
                                   c_void_p]),
    "heal_sp_wgrad_chunks": (c_int, [c_int]),
    "heal_sp_wgrad": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "heal_sp_gather_rows": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "heal_sp_table_capacity": (c_size_t, [c_int]),
    "heal_sp_hash_build": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    "heal_sp_neighbors": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                  c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    "heal_sp_out_sites_workspace": (c_size_t, [c_int, c_int]),
    "heal_sp_out_sites": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                  c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    "heal_conv_gemm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 8 + [c_void_p, c_void_p]),
    "heal_ln_stats": (c_int, [c_void_p, c_int, c_int, c_float, c_void_p, c_void_p]),
    "heal_linear": (c_int, [c_void_p, c_int, c_int, ctypes.c_longlong, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int,
                            c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                            ctypes.c_longlong, c_int, c_void_p]),
    "heal_split_attn_workspace": (c_size_t, [c_int, c_int, c_int]),
    "heal_split_attn_weights": (c_int, [c_void_p, ctypes.c_longlong, c_int, c_int, c_int] + [c_void_p] * 5 + [c_float] +
                                [c_void_p] * 5),
    "heal_split_attn_colsum": (c_int, [c_void_p, ctypes.c_longlong, c_int, c_int, c_int, c_void_p, c_void_p]),
    "heal_split_attn_weights_from_colsum": (c_int, [c_void_p, c_int, c_int, c_int, c_int] + [c_void_p] * 5 + [c_float] +
                                            [c_void_p] * 4),
    "heal_sp_rank_bytes": (c_size_t, [c_void_p, c_int]),
    "heal_sp_out_sites_rank": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                       c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    "heal_sp_neighbors_rank": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                       c_void_p, c_size_t, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "heal_sp_neighbors_root": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                       c_void_p, c_size_t, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "heal_sp_root_rank_bytes": (c_size_t, [c_void_p, c_int]),
    "heal_sp_pair_tiles_words": (c_size_t, [c_int, c_int]),
    "heal_sp_neighbor_tiles": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                       c_void_p, c_size_t, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "heal_sp_tiles_to_neighbors": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "heal_sp_conv_tiles_supported": (c_int, [c_int, c_int]),
    "heal_sp_conv_tiles": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                   c_void_p, c_void_p, c_void_p]),
    "heal_sp_root_rank": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                  c_size_t, c_void_p, c_void_p]),
    "heal_gconv_conv3_supported": (c_int, [c_int] * 5),
    "heal_gconv_conv3": (c_int, [c_void_p] * 6 + [c_int] * 7 + [c_void_p, c_void_p]),
    "heal_sp_transpose_neighbors": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "heal_sp_weight_fragments": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "heal_sp_conv": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                             c_void_p, c_void_p, c_void_p]),
    "heal_sp_to_bev_workspace": (c_size_t, [c_int, c_int, c_int, c_int]),
    "heal_sp_to_bev": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_size_t,
                               c_void_p, c_void_p]),
    "heal_agent_attention": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float,
                                     c_int, c_void_p, c_int, c_void_p]),
    "heal_agent_attention_backward": (c_int, [c_void_p] * 5 + [c_int, c_int, c_int, c_int, c_float, c_int] + [c_void_p] * 3 +
                                      [c_int, c_void_p]),
    "heal_grouped_conv3x3": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                     c_void_p, c_void_p]),
    "heal_bias_act": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "heal_resnext_bottleneck": (c_int, [c_void_p] * 7 + [c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "heal_upsample2x_bilinear": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "heal_depthwise_conv": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 11 + [c_void_p, c_void_p, c_void_p]),
    "heal_camera_matrices": (c_int, [c_void_p] * 5 + [c_int, c_void_p, c_void_p]),
    "heal_channel_dot": (c_int, [c_void_p] * 3 + [c_int] * 3 + [c_void_p, c_void_p]),
    "heal_layernorm_nchw": (c_int, [c_void_p] * 3 + [c_int] * 3 + [c_float, c_void_p, c_void_p]),
    "heal_se_gate": (c_int, [c_void_p] * 5 + [c_int] * 3 + [c_float, c_int, c_void_p, c_void_p]),
    "heal_conv1x1": (c_int, [c_void_p] * 5 + [c_int] * 8 + [c_void_p, c_void_p]),
    "heal_stem7x7": (c_int, [c_void_p, ctypes.c_longlong, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "heal_conv1x1_split_supported": (c_int, [c_int] * 4),
    "heal_conv1x1_split": (c_int, [c_void_p] * 4 + [c_int] * 7 + [c_void_p, c_void_p]),
    "heal_conv1x1_tiled_supported": (c_int, [c_int] * 5),
    "heal_conv1x1_tiled": (c_int, [c_void_p] * 4 + [c_int] * 9 + [c_void_p, c_void_p]),
    "heal_conv1x1_splitk_workspace": (c_size_t, [c_int] * 5),
    "heal_conv1x1_splitk": (c_int, [c_void_p] * 5 + [c_int] * 7 + [c_void_p, c_void_p, c_size_t, c_void_p]),
    "heal_conv1x1_d2s": (c_int, [c_void_p] * 3 + [c_int] * 9 + [c_void_p, c_void_p]),
    "heal_conv3x3": (c_int, [c_void_p] * 4 + [c_int] * 7 + [c_void_p, c_void_p]),
    "heal_conv3x3_same": (c_int, [c_void_p] * 3 + [c_int] * 11 + [c_void_p, c_void_p]),
    "heal_grouped16_conv3x3": (c_int, [c_void_p] * 3 + [c_int] * 5 + [c_void_p, c_void_p]),
    "heal_grouped_small_conv3x3": (c_int, [c_void_p] * 3 + [c_int] * 7 + [c_void_p, c_void_p]),
    "heal_conv3x3_winograd4": (c_int, [c_void_p] * 4 + [c_int] * 6 + [c_void_p, c_void_p]),
    "heal_conv3x3_winograd": (c_int, [c_void_p] * 4 + [c_int] * 7 + [c_void_p, c_void_p]),
    "heal_conv3x3_winograd_kc": (c_int, [c_void_p] * 4 + [c_int] * 8 + [c_void_p, c_void_p]),
    "heal_conv3x3_winograd_splitk_workspace": (c_size_t, [c_int] * 5),
    "heal_conv3x3_winograd_splitk": (c_int, [c_void_p] * 4 + [c_int] * 8 + [c_void_p, c_void_p, c_size_t, c_void_p]),
    "heal_nms_quads_workspace": (c_size_t, [c_int]),
    "heal_nms_quads": (c_int, [c_void_p, c_int, c_float, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    "heal_window_attention": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p,
                                      c_void_p]),
    "heal_window_attention_backward_workspace": (c_size_t, [c_int, c_int, c_int, c_int]),
    "heal_window_attention_backward": (c_int, [c_void_p] * 4 + [c_int] * 6 + [c_float, c_void_p, c_void_p, c_void_p, c_size_t,
                                                c_void_p]),
    "heal_label_assign_workspace": (c_size_t, [c_int]),
    "heal_label_assign": (c_int, [c_void_p, c_int, c_void_p, c_int, c_float, c_float, c_void_p, c_void_p, c_void_p,
                                  c_size_t, c_void_p]),
    "heal_boxes_bev_matrix": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "heal_nms_bev_workspace": (c_size_t, [c_int]),
    "heal_nms_bev": (c_int, [c_void_p, c_int, c_float, c_int, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
}


class HealAmdError(RuntimeError):
    pass


def declared_symbols(experimental=False):
    """Function names declared in include/heal_amd.h (the shipped C ABI), or with experimental=True in
    include/heal_amd_experimental.h (the measured-negative kernels of a HEAL_BUILD_EXPERIMENTAL=1 library)."""
    text = open(HEADER_EXPERIMENTAL if experimental else HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(heal_[a-z0-9_]+)\s*\(", text)))


def abi_version_of_header():
    m = re.search(r"#define\s+HEAL_AMD_ABI_VERSION\s+(\d+)", open(HEADER).read())
    if not m:
        raise HealAmdError(f"{HEADER} does not define HEAL_AMD_ABI_VERSION")
    return int(m.group(1))


def lib():
    """Load libheal_amd.so (raises if it has not been built: python -m heal_amd.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HealAmdError(
                f"{LIB_PATH} is missing: the HIP extension has not been built "
                "(run `python -m heal_amd.build`); heal_amd has no CPU fallback")
        # torch ships its own libamdhip64; it must be the HIP runtime of the process, so make sure it
        # is loaded before libheal_amd.so pulls in a second copy from /opt/rocm (two runtimes in one
        # process do not share devices, streams or allocations).
        import torch  # noqa: F401
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            if not hasattr(L, name):
                continue  # declared but not built yet -> surfaces in call() / the symbol test
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        # The argument lists above belong to ONE version of the C ABI: a stale library paired with newer Python (or the reverse)
        # would be called with the wrong arguments and corrupt memory silently.
        want = abi_version_of_header()
        got = int(L.heal_abi_version()) if hasattr(L, "heal_abi_version") else -1
        if got != want:
            raise HealAmdError(f"{LIB_PATH} has C-ABI version {got}, include/heal_amd.h declares {want}: rebuild it "
                               "(python -m heal_amd.build --force)")
        _lib = L
    return _lib


# HEAL_TRACE_CALLS=1: debugging aid for GPU memory faults (a fault kills the process without a Python traceback).  Every C-ABI call
# is announced on stderr BEFORE it is issued and the device is synchronised AFTER it (outside stream captures), so the last line
# printed names the faulting operator.  =2 also prints the pointer / integer arguments.
_TRACE = int(os.environ.get("HEAL_TRACE_CALLS", "0") or 0)


def _trace_call(name, args):
    import sys
    import torch
    cap = torch.cuda.is_current_stream_capturing()
    extra = ""
    if _TRACE >= 2:
        extra = " " + " ".join(hex(a.value or 0) if isinstance(a, ctypes.c_void_p) else str(a) for a in args
                               if isinstance(a, (int, float, ctypes.c_void_p)))
    print(f"[heal pid {os.getpid()}] {'capture ' if cap else ''}{name}{extra}", file=sys.stderr, flush=True)


def _trace_done(name):
    import sys
    import torch
    if not torch.cuda.is_current_stream_capturing():
        torch.cuda.synchronize()
        print(f"[heal pid {os.getpid()}]   ok {name}", file=sys.stderr, flush=True)


# ---- HEAL_GRAPH_GUARD=1: pointer ownership of captured graphs ---------------------------------------------------------
# A captured HIP graph has every address it was handed baked in; nothing in the runtime notices when one of them is later freed (ops.py
# keeps scratch and weight layouts alive by policy: "retire, never free").  In guard mode every device address passed through the C ABI
# DURING a capture is logged; whoever owns the graph takes the log (guard_take) and has it verified before replays (guard_check): each
# address must still lie in memory the caching allocator has handed out -- an active block of the ordinary pool, or any block of a
# graph-private pool that is still mapped (tensors allocated and released inside a capture live there for the life of their graph).
_GUARD = os.environ.get("HEAL_GRAPH_GUARD", "0") == "1"
_guard_log = []


def _segments():
    import torch
    segs = []
    for seg in torch.cuda.memory_snapshot():
        blocks, addr = [], seg["address"]
        for b in seg["blocks"]:
            a = b.get("address", addr)
            blocks.append((a, a + b["size"], b["state"]))
            addr = a + b["size"]
        pool = tuple(seg.get("segment_pool_id", (0, 0)))
        segs.append((seg["address"], seg["address"] + seg["total_size"], pool != (0, 0), blocks))
    segs.sort()
    return segs


def _locate(segs, addr):
    """-> None (no mapped segment: a host pointer or freed memory) | (private_pool, block_state)."""
    import bisect
    i = bisect.bisect_right(segs, (addr, float("inf"), True, [])) - 1
    if i < 0 or not (segs[i][0] <= addr < segs[i][1]):
        return None
    for lo, hi, state in segs[i][3]:
        if lo <= addr < hi:
            return segs[i][2], state
    return segs[i][2], "unknown"


def guard_begin():
    """Start of an OWNING capture (and every failure path of one): drop whatever earlier captures left in the log.  A capture that
    failed or was aborted never reaches guard_take(); without this its addresses -- tensors legitimately freed since -- would be
    inherited by the next owner and reported as 'freed' on its replays, and the log would grow without bound."""
    global _guard_log
    _guard_log = []


def guard_take():
    """The device addresses logged by captures since the last call: [(entry point, address)] (empty when the guard is off)."""
    global _guard_log
    log, _guard_log = _guard_log, []
    if not log:
        return []
    segs = _segments()
    seen, out = set(), []
    for name, addr in log:
        if addr not in seen and _locate(segs, addr) is not None:     # (host arrays also travel as void*: not guarded)
            seen.add(addr)
            out.append((name, addr))
    return out


def guard_check(entries, what="captured graph"):
    """Raise HealAmdError if any logged address no longer belongs to live allocator memory (see above).  One allocator snapshot."""
    if not entries:
        return
    segs = _segments()
    for name, addr in entries:
        where = _locate(segs, addr)
        if where is None:
            raise HealAmdError(f"{what}: {name} was captured with device address {addr:#x}, which is no longer mapped "
                               "(its tensor was freed and the memory returned to the driver)")
        private, state = where
        if not private and state != "active_allocated":
            raise HealAmdError(f"{what}: {name} was captured with device address {addr:#x}, whose block is now '{state}' "
                               "(its tensor was freed: a replay would read or overwrite somebody else's memory)")


def call(name, *args):
    """Call an int-returning entry point; raise HealAmdError with heal_last_error() on failure."""
    L = lib()
    if not hasattr(L, name):
        raise HealAmdError(f"libheal_amd.so does not export {name}")
    if _TRACE:
        _trace_call(name, args)
    if _GUARD:
        import torch
        if torch.cuda.is_current_stream_capturing():
            _guard_log.extend((name, a.value) for a in args if isinstance(a, ctypes.c_void_p) and a.value)
    rc = getattr(L, name)(*args)
    if rc != 0:
        raise HealAmdError(f"{name} failed: {L.heal_last_error().decode(errors='replace')}")
    if _TRACE:
        _trace_done(name)


def query(name, *args):
    """Call a size_t-returning workspace query."""
    L = lib()
    if not hasattr(L, name):
        raise HealAmdError(f"libheal_amd.so does not export {name}")
    return int(getattr(L, name)(*args))
