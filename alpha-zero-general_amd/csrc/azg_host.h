// host-side helpers shared by the two translation units of libazg_hip.so (azg.hip, azg_nn.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <string>

int azg_fail(const std::string& m);               // records the message for azg_last_error(), returns -1
static inline int fail(const std::string& m) { return azg_fail(m); }
#define HIPCHK(x) do { hipError_t _e = (x); if (_e != hipSuccess) return fail(std::string(#x) + ": " + hipGetErrorString(_e)); } while (0)


// internal (not part of include/azg.h): what azg_fused.hip needs from a forest handle owned by azg.hip
struct azg_forest;
namespace azg { struct ForestDev; }
const azg::ForestDev* azg_forest_dev_internal(azg_forest* f, int* game, int* variant, double* dirichlet_alpha);

// objects another translation unit hangs on a forest handle (the argument blocks / queues of the round kernels in azg_nn.hip): owned by
// the forest, released by azg_forest_destroy through `deleter` -- nothing is keyed by a forest's address, which a later forest can inherit
void azg_forest_attach(azg_forest* f, const char* key, void* obj, void (*deleter)(void*));
void* azg_forest_attached(azg_forest* f, const char* key);
