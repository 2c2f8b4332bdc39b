// Random-forest regression inference (the rf base learner of the stacked ensemble,
// Models/multi_input_data_regression_opt_transformer_cnn_20250113.py:262-266 and :394-403: RandomForestRegressor(300 trees,
// max_depth 30) over hstack([fingerprint, image]) = 49 319 features) for screening-scale batches.
// scikit-learn's arithmetic (sklearn/tree/_tree.pyx, ensemble/_forest.py), restated: X is float32; a node sends a sample
// left when (double)x[feature] <= threshold (thresholds are float64); a tree's prediction is the float64 value of the
// leaf; the forest's prediction is the sum over trees divided by the number of trees, in float64.
// The trees of a forest are concatenated into five node arrays; tree t owns nodes [root[t], root[t+1]).  One thread walks
// one sample through one group of trees (the walk is a chain of dependent, scattered loads: latency-bound, so many
// independent walks per CU), a second kernel adds the groups' partial sums in group order (deterministic).
#include "common.h"
#include "bbbp_hip.h"

namespace {

__global__ __launch_bounds__(256) void forest_walk_kernel(const float* X, long n, int n_features, const int* left, const int* right,
                                                         const int* feature, const double* threshold, const double* value,
                                                         const int* root, int n_trees, int groups, double* partial) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int g = blockIdx.y;
    if (i >= n) return;
    const float* x = X + i * (long)n_features;
    const int t0 = (int)((long)n_trees * g / groups), t1 = (int)((long)n_trees * (g + 1) / groups);
    double s = 0.0;
    for (int t = t0; t < t1; ++t) {
        int node = root[t];
        int l = left[node];
        while (l >= 0) {                                   // children_left == -1 marks a leaf
            node = (double)x[feature[node]] <= threshold[node] ? l : right[node];
            l = left[node];
        }
        s += value[node];
    }
    partial[(long)g * n + i] = s;
}

__global__ __launch_bounds__(256) void forest_reduce_kernel(const double* partial, long n, int groups, int n_trees, double* out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
    for (int g = 0; g < groups; ++g) s += partial[(long)g * n + i];
    out[i] = s / n_trees;
}

}  // namespace

extern "C" int bbbp_forest_groups(int n_trees) { return n_trees < 1 ? 0 : (n_trees < 32 ? n_trees : 32); }

extern "C" int bbbp_forest_predict(void* stream, const float* X, long n, int n_features, const int* left, const int* right,
                                   const int* feature, const double* threshold, const double* value, const int* root, int n_trees,
                                   double* partial, double* out) {
    BBBP_CHECK_ARG(n >= 0 && n_features >= 1 && n_trees >= 1, "forest_predict: bad sizes");
    if (n == 0) return BBBP_OK;
    BBBP_CHECK_ARG(X && left && right && feature && threshold && value && root && partial && out, "forest_predict: null pointer");
    const int groups = bbbp_forest_groups(n_trees);
    const long blocks = (n + 255) / 256;
    BBBP_CHECK_ARG(blocks <= 0x7fffffffL, "forest_predict: too many rows");
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(forest_walk_kernel, dim3((unsigned)blocks, groups), dim3(256), 0, st, X, n, n_features, left, right, feature, threshold,
                       value, root, n_trees, groups, partial);
    BBBP_CHECK_LAUNCH();
    hipLaunchKernelGGL(forest_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, partial, n, groups, n_trees, out);
    BBBP_CHECK_LAUNCH();
    return BBBP_OK;
}

// ------------------------------------------------------------------------------------------------------------------------------
// Gradient-boosted regression trees, prediction only (the xgb base learner of the stack: XGBRegressor(300 trees, max_depth 30,
// tree_method="hist") over hstack([fingerprint, image]), Models/multi_input_data_regression_opt_transformer_cnn_20250108.py:186-189;
// its fitted form ships as Models/xgb_model_maccs.pkl).  XGBoost's published predict rule (src/predictor/cpu_predictor.cc, RegTree::
// GetNext), restated: a node sends a row to its left child when x[split_index] < split_condition (float32 compare), to the default
// child (default_left) when the feature is missing (NaN); a leaf's value sits in split_conditions[leaf]; the margin is
// base_score + (((l_0 + l_1) + l_2) + ...) summed in float32 in tree order; reg:squarederror returns the margin.
// The xgboost package is absent from this image: parity of this path is UNPINNED (oracle = the same rule restated in numpy).
// Kernel 1 walks every (tree, row) pair in parallel into leaf[tree][row]; kernel 2 adds a row's leaves in tree order.
#define BBBP_GBT_MAX_DEPTH 1024
namespace {
__global__ __launch_bounds__(256) void gbt_walk_kernel(const float* X, long n, int n_features, const int* left, const int* right,
                                                      const int* feature, const float* cond, const uint8_t* default_left, const int* root,
                                                      float* leaf) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int t = blockIdx.y;
    if (i >= n) return;
    const float* x = X + i * (long)n_features;
    int node = root[t];
    int l = left[node];
    // the host validates the trees (boosters.validate_gbt: children inside the tree, one parent per node, depth <= 1024); the walk is
    // bounded all the same, so that arrays handed to the C ABI directly cannot spin a wave for ever: NaN marks a walk that gave up
    int steps = 0;
    while (l >= 0 && steps < BBBP_GBT_MAX_DEPTH) {
        const float v = x[feature[node]];
        node = (v != v) ? (default_left[node] ? l : right[node]) : (v < cond[node] ? l : right[node]);
        l = left[node];
        ++steps;
    }
    leaf[(long)t * n + i] = (l >= 0) ? __builtin_nanf("") : cond[node];
}
__global__ __launch_bounds__(256) void gbt_sum_kernel(const float* leaf, long n, int n_trees, float base_score, float* out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int t = 0; t < n_trees; ++t) s += leaf[(long)t * n + i];
    out[i] = base_score + s;
}
}  // namespace

extern "C" int bbbp_gbt_predict(void* stream, const float* X, long n, int n_features, const int* left, const int* right, const int* feature,
                                const float* split_condition, const uint8_t* default_left, const int* root, int n_trees, float base_score,
                                float* leaf_scratch, float* out) {
    BBBP_CHECK_ARG(n >= 0 && n_features >= 1 && n_trees >= 1 && n_trees <= 65535, "gbt_predict: bad sizes (n_trees 1..65535)");
    if (n == 0) return BBBP_OK;
    BBBP_CHECK_ARG(X && left && right && feature && split_condition && default_left && root && leaf_scratch && out, "gbt_predict: null pointer");
    const long blocks = (n + 255) / 256;
    BBBP_CHECK_ARG(blocks <= 0x7fffffffL, "gbt_predict: too many rows");
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(gbt_walk_kernel, dim3((unsigned)blocks, n_trees), dim3(256), 0, st, X, n, n_features, left, right, feature,
                       split_condition, default_left, root, leaf_scratch);
    BBBP_CHECK_LAUNCH();
    hipLaunchKernelGGL(gbt_sum_kernel, dim3((unsigned)blocks), dim3(256), 0, st, leaf_scratch, n, n_trees, base_score, out);
    BBBP_CHECK_LAUNCH();
    return BBBP_OK;
}

// ------------------------------------------------------------------------------------------------------------------------------
// Oblivious (symmetric) trees, prediction only: the cat base learner of the stack (CatBoostRegressor(iterations=300, depth=10),
// Models/multi_input_data_regression_opt_transformer_cnn_20250108.py:192-195).  CatBoost's published rule for float features
// (catboost/libs/model: the JSON export's "oblivious_trees"): level i of a tree compares ONE (feature, border) pair for every row,
// bit i of the leaf index is x[feature] > border (a NaN compares false, or true for features whose nan_value_treatment is "AsTrue"),
// the tree adds leaf_values[index]; the prediction is scale * (sum over trees, float64, tree order) + bias.
// The catboost package is absent from this image and the reference ships no fitted CatBoost model: parity UNPINNED.
namespace {
__global__ __launch_bounds__(256) void obl_walk_kernel(const float* X, long n, int n_features, const int* split_feature, const float* split_border,
                                                      const uint8_t* nan_true, const int* tree_first_split, const long* tree_first_leaf,
                                                      const double* leaf_values, double* leaf) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int t = blockIdx.y;
    if (i >= n) return;
    const float* x = X + i * (long)n_features;
    const int s0 = tree_first_split[t], s1 = tree_first_split[t + 1];
    unsigned idx = 0;
    for (int s = s0; s < s1; ++s) {
        const int f = split_feature[s];
        const float v = x[f];
        const bool bit = (v != v) ? (nan_true[f] != 0) : (v > split_border[s]);
        idx |= (bit ? 1u : 0u) << (s - s0);
    }
    leaf[(long)t * n + i] = leaf_values[tree_first_leaf[t] + idx];
}
__global__ __launch_bounds__(256) void obl_sum_kernel(const double* leaf, long n, int n_trees, double scale, double bias, double* out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
    for (int t = 0; t < n_trees; ++t) s += leaf[(long)t * n + i];
    {
#pragma clang fp contract(off)                               // two roundings, as the host-side rule (no fused multiply-add)
        const double prod = scale * s;
        out[i] = prod + bias;
    }
}
}  // namespace

extern "C" int bbbp_oblivious_predict(void* stream, const float* X, long n, int n_features, const int* split_feature, const float* split_border,
                                      const uint8_t* nan_true, const int* tree_first_split, const long* tree_first_leaf, const double* leaf_values,
                                      int n_trees, double scale, double bias, double* leaf_scratch, double* out) {
    BBBP_CHECK_ARG(n >= 0 && n_features >= 1 && n_trees >= 1 && n_trees <= 65535, "oblivious_predict: bad sizes (n_trees 1..65535)");
    if (n == 0) return BBBP_OK;
    BBBP_CHECK_ARG(X && split_feature && split_border && nan_true && tree_first_split && tree_first_leaf && leaf_values && leaf_scratch && out,
                   "oblivious_predict: null pointer");
    const long blocks = (n + 255) / 256;
    BBBP_CHECK_ARG(blocks <= 0x7fffffffL, "oblivious_predict: too many rows");
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(obl_walk_kernel, dim3((unsigned)blocks, n_trees), dim3(256), 0, st, X, n, n_features, split_feature, split_border, nan_true,
                       tree_first_split, tree_first_leaf, leaf_values, leaf_scratch);
    BBBP_CHECK_LAUNCH();
    hipLaunchKernelGGL(obl_sum_kernel, dim3((unsigned)blocks), dim3(256), 0, st, leaf_scratch, n, n_trees, scale, bias, out);
    BBBP_CHECK_LAUNCH();
    return BBBP_OK;
}
