"""ctypes binding of libstin_hip.so (the C ABI declared in include/stin_hip.h).

There is NO fallback: if the shared library is missing or a symbol is absent the
import of the HIP path fails loudly (``StinLibraryError``).  Build it with
``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C surface_texture_inpainting_net_amd/csrc``.
"""
import ctypes
import os

# PyTorch-ROCm bundles its own HIP runtime (torch/lib/libamdhip64.so, SONAME libamdhip64.so.7).  It MUST
# be in the process before libstin_hip.so is dlopen'ed so that the library's NEEDED libamdhip64.so.7
# resolves to that same runtime instance; loading /opt/rocm's copy next to torch's gives a second runtime
# that sees no device (hipErrorNoDevice on the first launch).
import torch  # noqa: F401  (side effect: loads the HIP runtime torch uses)

_HERE = os.path.dirname(os.path.abspath(__file__))
# (STIN_LIB_PATH: an alternative build of the same library, for same-box A/B runs of compile-time switches)
LIB_PATH = os.environ.get('STIN_LIB_PATH') or os.path.join(_HERE, 'libstin_hip.so')

c_i64, c_i32, c_int, c_f32, c_f64 = ctypes.c_int64, ctypes.c_int32, ctypes.c_int, ctypes.c_float, ctypes.c_double
c_ptr, c_size = ctypes.c_void_p, ctypes.c_size_t


class StinLibraryError(RuntimeError):
    pass


class StinError(RuntimeError):
    pass


# name -> (restype, argtypes); mirrors include/stin_hip.h one to one
SIGNATURES = {
    'stin_version': (c_int, []),
    'stin_error_string': (ctypes.c_char_p, [c_int]),
    'stin_csr_workspace_bytes': (c_size, [c_i64, c_i64]),
    'stin_csr_from_coo_i64': (c_int, [c_ptr, c_ptr, c_i64, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr,
                                      c_size, c_ptr]),
    'stin_csr_pair_from_edges_i64': (c_int, [c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr,
                                             c_ptr, c_ptr, c_size, c_ptr]),
    'stin_narrow_i64_to_i32': (c_int, [c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_ptr]),
    'stin_segment_sum_f32': (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_int, c_int, c_ptr, c_i64, c_ptr]),
    'stin_edge_relu_mean_fwd_f32': (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_int, c_ptr, c_i64,
                                            c_int, c_ptr, c_ptr]),
    'stin_edge_relu_mean_bwd_dst_mask_f32': (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_int, c_ptr, c_i64, c_ptr]),
    'stin_edge_relu_mean_bwd_src_mask_f32': (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_ptr,
                                                     c_i64, c_ptr]),
    'stin_edge_relu_mean_bwd_mask_f32': (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_ptr,
                                                 c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_int, c_ptr]),
    'stin_edge_relu_mean_bwd_dst_f32': (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_int,
                                                c_ptr, c_i64, c_ptr]),
    'stin_edge_relu_mean_bwd_src_f32': (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_i64,
                                                c_int, c_ptr, c_i64, c_ptr]),
    'stin_pool_max_fwd_f32': (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_int, c_ptr, c_i64, c_ptr, c_ptr]),
    'stin_pool_max_bwd_f32': (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_int, c_ptr, c_i64, c_ptr]),
    'stin_gather_rows_f32': (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_int, c_ptr, c_i64, c_ptr]),
    'stin_batch_pool_i64': (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_ptr]),
    'stin_norm_group_ids_i64': (c_int, [c_ptr, c_ptr, c_int, c_i64, c_ptr, c_ptr, c_ptr]),
    'stin_gather_i64': (c_int, [c_ptr, c_ptr, c_i64, c_ptr, c_ptr]),
    'stin_colreduce_workspace_bytes': (c_size, [c_int, c_int]),
    'stin_colreduce_f32': (c_int, [c_int, c_ptr, c_i64, c_ptr, c_i64, c_i64, c_int, c_ptr, c_int, c_ptr, c_ptr, c_ptr,
                                   c_ptr, c_ptr, c_int, c_ptr, c_f32, c_ptr, c_ptr, c_ptr, c_size, c_ptr]),
    'stin_norm_act_res_fwd_f32': (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_int, c_int, c_ptr,
                                          c_i64, c_ptr]),
    'stin_norm_act_bwd_f32': (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr,
                                      c_i64, c_int, c_int, c_ptr, c_i64, c_ptr]),
    'stin_edge_relu_mean_fwd_ti_f32': (c_int, [c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_int, c_ptr, c_i64, c_int, c_ptr, c_ptr]),
    'stin_edge_bwd_ti_colsum_rows': (c_i64, [c_i64, c_int]),
    'stin_edge_bwd_ti_colsum_fold_f32': (c_int, [c_ptr, c_i64, c_int, c_ptr, c_ptr]),
    'stin_edge_relu_mean_bwd_mask_ti_f32': (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_ptr, c_i64,
                                                    c_ptr, c_i64, c_ptr, c_i64, c_int, c_ptr, c_i64, c_ptr]),
    'stin_gemm_nt_f32': (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_ptr, c_i64, c_i64, c_int, c_int,
                                 c_ptr, c_i64, c_int, c_ptr]),
    'stin_gemm_tn_workspace_bytes': (c_size, [c_i64, c_int, c_int, c_int]),
    'stin_gemm_tn_f32': (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_i64, c_int, c_int, c_int, c_ptr, c_i64, c_ptr, c_i64,
                                 c_int, c_ptr, c_size, c_ptr]),
    'stin_gemm_tn_wb_f32': (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_i64, c_int, c_int, c_ptr, c_i64, c_ptr, c_i64, c_ptr,
                                    c_int, c_ptr, c_size, c_ptr]),
    'stin_gemm_tn_wb_bf16': (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_i64, c_int, c_int, c_ptr, c_i64, c_ptr, c_i64, c_ptr,
                                     c_ptr, c_size, c_ptr]),
    'stin_cols_axpy_rowmask_f32': (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_int, c_int, c_f32, c_ptr]),
    'stin_cols_axpy_rowmask_bf16': (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_int, c_int, c_f32, c_ptr]),
    'stin_voxel_cluster_workspace_bytes': (c_size, [c_i64]),
    'stin_voxel_cluster_f64': (c_int, [c_ptr, c_i64, c_f64, c_ptr, c_ptr, c_ptr, c_ptr, c_size, c_ptr]),
    'stin_coalesce_workspace_bytes': (c_size, [c_i64]),
    'stin_coalesce_pairs_i64': (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_size, c_ptr]),
    'stin_gemm_nt_dotelu_groups': (c_i64, [c_i64, c_int, c_int, c_int]),
    'stin_gemm_nt_dotelu_f32': (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_i64, c_int, c_int, c_ptr, c_i64, c_int,
                                        c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_size, c_ptr]),
    'stin_norm_coef_from_partials_f32': (c_int, [c_ptr, c_i64, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    'stin_pad_rows_f32': (c_int, [c_ptr, c_i64, c_i64, c_int, c_int, c_ptr, c_ptr]),
    'stin_pad_rows_bf16': (c_int, [c_ptr, c_i64, c_i64, c_int, c_int, c_ptr, c_ptr]),
    'stin_linear_tanh_bwd_workspace_bytes': (c_size, [c_i64, c_int, c_int]),
    'stin_linear_tanh_fwd_f32': (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_int, c_int, c_ptr, c_ptr]),
    'stin_linear_tanh_fwd_bf16': (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_int, c_int, c_ptr, c_ptr]),
    'stin_linear_tanh_bwd_f32': (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_i64, c_int, c_int, c_ptr, c_i64, c_ptr, c_ptr,
                                         c_ptr, c_size, c_ptr]),
    'stin_linear_tanh_bwd_bf16': (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_i64, c_int, c_int, c_ptr, c_i64, c_ptr, c_ptr,
                                          c_ptr, c_size, c_ptr]),
    'stin_edgeconv_pack_f32': (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_int, c_int, c_ptr,
                                       c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_ptr]),
    'stin_gemm_split_weights_f32': (c_int, [c_ptr, c_i64, c_int, c_int, c_int, c_ptr, c_i64, c_ptr]),
    'stin_gemm_w_is_frag': (c_int, [c_int, c_int]),
    'stin_edgeconv_unpack_grads_f32': (c_int, [c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_int, c_int, c_ptr, c_ptr, c_ptr,
                                               c_ptr, c_ptr, c_ptr, c_ptr]),
    'stin_norm_bwd_coef_f32': (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_ptr, c_ptr, c_ptr]),
    'stin_masked_l1_workspace_bytes': (c_size, [c_i64, c_int]),
    'stin_masked_l1_loss_f32': (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_ptr, c_ptr, c_ptr, c_size, c_ptr]),
    'stin_total_variation_workspace_bytes': (c_size, [c_i64]),
    'stin_total_variation_f32': (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_int, c_ptr, c_ptr, c_size, c_ptr]),
    'stin_graph_laplace_f32': (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_int, c_ptr, c_i64, c_ptr]),
    'stin_adam_f32': (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_f64, c_f64, c_f64, c_f64, c_f64, c_int, c_int,
                              c_ptr]),
}
# bf16-storage variants: same argument lists as their *_f32 twins (pointers are void* here)
for _n in ('stin_segment_sum', 'stin_edge_relu_mean_fwd', 'stin_edge_relu_mean_bwd_dst_mask',
           'stin_edge_relu_mean_bwd_src_mask', 'stin_edge_relu_mean_bwd_mask', 'stin_pool_max_fwd', 'stin_pool_max_bwd', 'stin_gather_rows',
           'stin_colreduce', 'stin_norm_act_res_fwd', 'stin_norm_act_bwd'):
    SIGNATURES[_n + '_bf16'] = SIGNATURES[_n + '_f32']
SIGNATURES['stin_gemm_nt_bf16'] = SIGNATURES['stin_gemm_nt_f32']          # last int = c_is_f32 instead of precision
SIGNATURES['stin_gemm_tn_bf16'] = (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_i64, c_int, c_int, c_int, c_ptr, c_i64, c_ptr,
                                           c_i64, c_ptr, c_size, c_ptr])
for _n in ('stin_dilated_walk_f32', 'stin_dilated_walk_f64'):
    SIGNATURES[_n] = (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, ctypes.POINTER(c_i32), c_int, c_ptr, c_ptr])
SIGNATURES['stin_edgeconv_block_fwd_workspace_bytes'] = (c_size, [c_int] * 6)
SIGNATURES['stin_edgeconv_block_fwd'] = (c_int, [c_int, c_ptr, c_i64, c_i64] + [c_int] * 6 + [c_ptr] * 6 + [c_ptr] * 3 + [c_int, c_ptr, c_ptr,
                                                 c_int, c_f32, c_int, c_int, c_int, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr,
                                                 c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_size, c_ptr])
SIGNATURES['stin_gemm_nt_colstats_groups'] = (c_i64, [c_i64, c_int, c_int, c_int])
SIGNATURES['stin_gemm_nt_colstats_f32'] = (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_ptr, c_i64, c_i64, c_int, c_int,
                                                   c_ptr, c_i64, c_int, c_ptr, c_size, c_ptr])
SIGNATURES['stin_moments_final_f32'] = (c_int, [c_ptr, c_i64, c_int, c_ptr, c_f32, c_ptr, c_ptr, c_ptr])
SIGNATURES['stin_norm_fold_rows'] = (c_int, [c_i64, c_int, c_i64])
SIGNATURES['stin_norm_act_res_fwd_fold_f32'] = (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_f32, c_i64, c_int, c_ptr, c_ptr,
                                                        c_ptr, c_i64, c_ptr])
SIGNATURES['stin_norm_act_bwd_fold_f32'] = (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_ptr,
                                                    c_i64, c_ptr])
SIGNATURES['stin_edgeconv_block_fwd_pack_offsets'] = (c_int, [c_int, c_int, c_int, c_int, c_ptr, c_ptr, c_ptr])
SIGNATURES['stin_edgeconv_pack_many_f32'] = (c_int, [c_ptr, c_int, c_i64, c_ptr])
SIGNATURES['stin_edgeconv_block_bwd_workspace_bytes'] = (c_size, [c_i64, c_int, c_int, c_int, c_int, c_int, c_int])
SIGNATURES['stin_edgeconv_block_bwd'] = (c_int, [c_int, c_ptr, c_i64, c_ptr, c_i64, c_i64] + [c_int] * 6 + [c_ptr, c_i64, c_ptr, c_i64,
                                                 c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr] + [c_ptr] * 6 + [c_int, c_ptr, c_ptr, c_ptr, c_int,
                                                 c_int, c_ptr, c_i64] + [c_ptr] * 6 + [c_ptr, c_size, c_ptr] + [c_ptr] * 4 + [c_int])
SIGNATURES['stin_edgeconv_chain_fwd'] = (c_int, [c_int, c_ptr, c_int, c_ptr, c_i64, c_i64, c_int, c_int, c_int, c_ptr, c_int, c_ptr, c_ptr,
                                                 c_int, c_f32, c_size, c_ptr])
SIGNATURES['stin_edgeconv_chain_bwd'] = (c_int, [c_int, c_ptr, c_int, c_ptr, c_i64, c_ptr, c_i64, c_i64, c_int, c_int, c_int, c_ptr, c_int,
                                                 c_ptr, c_ptr, c_ptr, c_int, c_ptr, c_i64, c_ptr, c_ptr, c_size, c_ptr, c_ptr])
SIGNATURES['stin_vertex_order_workspace_bytes'] = (c_size, [c_i64])
SIGNATURES['stin_vertex_order_f32'] = (c_int, [c_ptr, c_i64, c_ptr, c_int, c_ptr, c_size, c_ptr])
SIGNATURES['stin_relabel_many_i64'] = (c_int, [c_ptr, c_int, c_ptr])
SIGNATURES['stin_net_fwd'] = (c_int, [c_int, c_ptr, c_int, c_ptr])
SIGNATURES['stin_net_bwd'] = (c_int, [c_int, c_ptr, c_int, c_ptr, c_i64, c_int, c_ptr, c_ptr])
SIGNATURES['stin_plan_build_workspace_bytes'] = (c_size, [c_i64, c_i64])
SIGNATURES['stin_plan_build_many'] = (c_int, [c_ptr, c_int, c_ptr, c_ptr, c_size, c_ptr])
SIGNATURES['stin_edgeconv_wgrad_workspace_bytes'] = (c_size, [c_i64, c_int, c_int, c_int, c_int])
SIGNATURES['stin_edgeconv_wgrad'] = (c_int, [c_int, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_i64] + [c_int] * 7 + [c_ptr] * 6 +
                                     [c_ptr, c_size, c_ptr])
SIGNATURES['stin_edgeconv_wgrad_ti'] = (c_int, [c_int, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_i64] + [c_int] * 7 + [c_ptr] * 6 +
                                        [c_ptr, c_i64, c_ptr, c_size, c_ptr])
SIGNATURES['stin_norm_bwd_coef_m_quirk_f32'] = (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_ptr, c_ptr])
SIGNATURES['stin_gather_add_rows_f32'] = (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_ptr, c_i64, c_int, c_ptr, c_i64, c_ptr])
SIGNATURES['stin_segment_mean_stats_groups'] = (c_i64, [c_i64, c_int])
SIGNATURES['stin_segment_mean_stats_f32'] = (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_int, c_ptr, c_i64, c_ptr, c_size, c_ptr])
SIGNATURES['stin_gather_add_rows_stats_groups'] = (c_i64, [c_i64, c_int])
SIGNATURES['stin_gather_add_rows_stats_f32'] = (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_ptr, c_i64, c_int, c_ptr, c_i64, c_ptr, c_size, c_ptr])
SIGNATURES['stin_bn_mean_bwd_f32'] = (c_int, [c_ptr, c_i64, c_ptr, c_i64] + [c_ptr] * 7 + [c_f32, c_i64, c_int, c_ptr, c_i64, c_ptr])
SIGNATURES['stin_bn_running_stats_f32'] = (c_int, [c_ptr, c_ptr, c_int, c_f32, c_f32, c_f32, c_ptr, c_ptr, c_ptr])
SIGNATURES['stin_gemm_nt_bn_f32'] = (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_ptr, c_i64, c_int, c_ptr])
SIGNATURES['stin_gemm_tn_bn_f32'] = (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_ptr, c_i64, c_int,
                                             c_ptr, c_size, c_ptr])
SIGNATURES['stin_gemm_nt_stream_f32'] = (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_ptr, c_i64, c_int, c_ptr])
SIGNATURES['stin_gemm_nt_bn_bwd_groups'] = (c_i64, [c_i64, c_int, c_int, c_int])
SIGNATURES['stin_gemm_nt_bn_bwd_stats_f32'] = (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int,
                                                       c_int, c_ptr, c_size, c_ptr, c_ptr])
SIGNATURES['stin_gemm_nt_bn_bwd_apply_f32'] = (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_f32, c_i64,
                                                       c_int, c_int, c_ptr, c_i64, c_int, c_ptr])
SIGNATURES['stin_scmn_pack_f32'] = (c_int, [c_ptr] * 6 + [c_int] * 4 + [c_ptr] * 6)
SIGNATURES['stin_scmn_unpack_f32'] = (c_int, [c_ptr, c_int, c_int, c_int, c_ptr, c_ptr])
SIGNATURES['stin_bn_affine_res_fwd_f32'] = (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_int, c_int, c_ptr,
                                                    c_i64, c_ptr])
SIGNATURES['stin_relu_mask_bwd_f32'] = (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_int, c_int, c_ptr, c_ptr, c_ptr])
SIGNATURES['stin_concat_unpool_f32'] = (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_int, c_int, c_ptr, c_i64, c_ptr])
SIGNATURES['stin_bn_act_fwd_f32'] = (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_ptr, c_i64, c_ptr])
SIGNATURES['stin_bn_act_bwd_f32'] = (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_f32, c_i64, c_int,
                                             c_int, c_ptr, c_i64, c_ptr])

_lib = None


def load():
    """-> the ctypes CDLL with argtypes/restype set for every declared symbol."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise StinLibraryError(
            'libstin_hip.so not found at %s - the HIP extension must be built (no CPU/eager fallback exists): '
            'run `make -C %s`' % (LIB_PATH, os.path.join(_HERE, 'csrc')))
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise StinLibraryError('cannot load %s: %s' % (LIB_PATH, e))
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise StinLibraryError('libstin_hip.so lacks symbol %s (stale build?)' % name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code, what=''):
    if code != 0:
        msg = load().stin_error_string(int(code)).decode()
        raise StinError('%s failed: %s (code %d)' % (what or 'stin call', msg, code))
