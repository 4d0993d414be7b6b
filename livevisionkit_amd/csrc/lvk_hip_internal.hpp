// Internal definitions shared by the translation units of liblvk_hip.so (not installed).
#pragma once

#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>
#include <map>
#include <utility>

#include "lvk_hip.h"

struct LinTabEntry { int s0, s1; float a0, a1; };   // one column/row of the INTER_LINEAR mesh->frame table

struct lvk_hip_ctx
{
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    std::string last_error;

    // Pinned staging ring for small host->device parameter blocks (meshes, tables).
    static constexpr int kStageSlots = 16;
    static constexpr size_t kStageBytes = 64 * 1024;
    uint8_t* stage_host = nullptr;      // kStageSlots * kStageBytes, pinned
    uint8_t* stage_dev = nullptr;       // same size, device
    hipEvent_t stage_done[kStageSlots] = {};
    int stage_next = 0;

    // Cached INTER_LINEAR tables: key = (mesh extent, frame extent, vertical?)
    std::map<std::tuple<int, int, int>, LinTabEntry*> lintabs;

    int fail(int code, const std::string& msg) { last_error = msg; return code; }
};

#define LVK_HIP_CHECK(ctx, expr)                                                                      \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess)                                                                         \
            return (ctx)->fail(LVK_HIP_ERR_RUNTIME, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

#define LVK_HIP_REQUIRE(ctx, cond)                                                                    \
    do { if (!(cond)) return (ctx)->fail(LVK_HIP_ERR_ARG, "pre-condition failed: " #cond); } while (0)

// Copies `bytes` (<= kStageBytes) of host data into a device staging slot, asynchronously on the
// context's stream, and returns the device address.  The slot is recycled after kStageSlots uses.
int lvk_stage_params(lvk_hip_ctx* ctx, const void* host, size_t bytes, void** d_out);

// Device-resident INTER_LINEAR table for resizing a mesh axis of `msize` vertices to `fsize` pixels.
int lvk_get_lintab(lvk_hip_ctx* ctx, int msize, int fsize, bool vertical, const LinTabEntry** d_out);
