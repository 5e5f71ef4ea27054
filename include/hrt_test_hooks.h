/* hrt_test_hooks.h -- entry points that exist ONLY in libhip_raytrace_test.so (the same sources as libhip_raytrace.so compiled
 * with -DHRT_TEST_HOOKS): what the test suite needs beyond the product ABI of hip_raytrace.h.  A production host never links
 * against them; `nm -D libhip_raytrace.so` shows none.
 * They probe the arithmetic contract the kernels share with the oracle (XMath call sites: Engine/Float3.cs:69-106,
 * Engine/RTRay.cs:570-653, Engine/SceneDeviceViews.cs:500-546) and expose host-side derived structures (second TLAS,
 * treelets) for structural tests. */
#ifndef HRT_TEST_HOOKS_H
#define HRT_TEST_HOOKS_H
#include "hip_raytrace.h"
#ifdef __cplusplus
extern "C" {
#endif

/* test hook: evaluates function `fn` of include/hrt_math.h on device slot 0 for n inputs
 * (fn ids as in tests/test_math_gpu.py); lets the GPU tests check bit-equality of the
 * arithmetic contract against the oracle.  Not needed by a production host. */
int  hrt_math_probe(hrt_ctx* ctx, int fn, int n, const float* x, const float* y, float* out);

/* test hook: compares a trimmed device-side function with its IEEE definition for EVERY float of its stated domain, on the
 * device (which: 0 = 1/sqrt(x) of Normalize for x in [1e-20, +inf], 1 = the square root of the hemisphere sampler for +0 and
 * [2^-96, +inf]).  *mismatches = number of differing results (0 expected), *first_bad (may be NULL) = bits of the smallest one. */
int  hrt_math_exhaustive(hrt_ctx* ctx, int which, uint64_t* mismatches, uint32_t* first_bad);

/* test hooks, host code only (no device, no context): what hrt_scene_upload computes on the host for the SECOND tree of a scene of
 * many one-sphere instances.
 * hrt_debug_second_tree_topology: the binned-SAH topology over the world bounds of n instances, in walk order.  order[n]: instance
 *   of every leaf slot; per node i < *n_nodes (arrays of capacity 2 n): link[i] = first slot of a leaf / index of the first child,
 *   skip[i] = next node when node i is missed (0x0FFFFFFF: none), count[i] = instances of a leaf (0: inner node), parent[i].
 * hrt_debug_second_tree_reorder: the renumbering of a node array (records of 8 floats: lo.xyz, link word, hi.xyz, skip word with
 *   the count in its top four bits; inlined != 0: every leaf is followed by one record per instance, count field 15) for rays
 *   whose direction has the signs sign[3] (+1 / -1 / 0: builder's order along that axis); links of the result are offset by base;
 *   from[i] = record of the input at position i.  Returns 0, or HRT_ERR_INVALID_ARG if the input is not such a tree. */
int  hrt_debug_second_tree_topology(const hrt_instance* instances, int32_t n, int32_t* order, int32_t* link, int32_t* skip,
                                    int32_t* count, int32_t* parent, int32_t* n_nodes);
int  hrt_debug_second_tree_reorder(const float* records, int32_t n_records, const int32_t* sign, int32_t base, int32_t inlined,
                                   float* out_records, int32_t* from);

/* Treelets (csrc/hrt_treelets.hpp; the walker of HRT_FLAG_TREELETS).
 * hrt_debug_set_treelet_limits: limits of the treelet cut for scenes committed from now on, process-wide: LDS bytes of a treelet,
 *   smallest subtree that becomes one, smallest BLAS that gets any (bytes <= 0: the shipped values; min_blas_nodes < 0: none at all).
 * hrt_debug_treelet_count: treelets of the committed scene on device slot 0 (0: none, or dropped by a vertex update).
 * hrt_debug_treelets: host code only.  Runs the upload's validation + repack + treelet cut on a scene and returns the packed BLAS
 *   node array (records of 8 floats: lo.xyz, link word, hi.xyz, skip | count << 28), the reduced trees (same records, explicit
 *   links; red_orig[i] = packed index of the node a reduced record copies), the treelet table (8 ints per treelet: nodeLo, nodeHi,
 *   triLo, triHi, exitRed, 0, 0, 0) and red_of_root[n blas nodes].  counts[3] = {blas nodes, reduced records, treelets}; arrays may
 *   be NULL to query the counts.  Returns 0 or an hrt_status. */
int  hrt_debug_set_treelet_limits(int bytes, int min_nodes, int min_blas_nodes);
int  hrt_debug_treelet_count(hrt_ctx* ctx);
int  hrt_debug_treelets(const hrt_scene_desc* scene, int bytes, int min_nodes, int min_blas_nodes,
                        float* blas, float* red, int32_t* red_orig, int32_t* treelets, int32_t* red_of_root, int64_t* counts);

#ifdef __cplusplus
}
#endif
#endif
