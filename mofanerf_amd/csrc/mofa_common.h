// Shared device/host helpers for libmofanerf_hip.so (gfx950 only — no portability layer).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/mofanerf_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace mofa {

constexpr int kRowTile = MOFA_ROW_TILE;  // activation rows per workgroup tile
constexpr int kPanelK = 16;              // floats per panel row (64 B)

__host__ __device__ inline int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

// element (row,k) of a panel matrix with `rows` rows (see include/mofanerf_hip.h, "Panel layout")
__host__ __device__ inline int64_t panel_index(int64_t rows, int64_t row, int k) {
    return (int64_t)(k >> 4) * rows * kPanelK + row * kPanelK + ((((k >> 2) & 3) ^ ((int)(row >> 2) & 3)) << 2) +
           (k & 3);
}

// ReLU with torch's NaN behaviour (F.relu(nan) = nan; fmaxf(nan, 0) would be 0): a NaN that enters the network - bad
// input, or an fp16 overflow in the opt-in split mode - must reach the image exactly as it does in the reference.
__device__ __forceinline__ float relu_np(float v) { return v < 0.f ? 0.f : v; }

void set_error(const char* fmt, ...);
int check_launch(const char* what);

}  // namespace mofa

#define MOFA_REQUIRE(cond, ...)              \
    do {                                     \
        if (!(cond)) {                       \
            mofa::set_error(__VA_ARGS__);    \
            return MOFA_EINVAL;              \
        }                                    \
    } while (0)
