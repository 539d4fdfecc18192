// pinn_inst.h -- per-width launch entry points (one translation unit per padded hidden width HP).
#pragma once
#include "pinn_kernel.h"

// returns 0 on success, 1 if (nd, n2) has no instantiation, 2 on launch failure.
// `query` != 0: do not launch; write LDS bytes / threads / slab vec4 per WG / suggested WGs per CU to info[0..3].
int pinn_launch_tile_hp16(int nd, int n2, const PinnKArgs* a, int grid, void* stream, int query, long long* info);
int pinn_launch_tile_hp32(int nd, int n2, const PinnKArgs* a, int grid, void* stream, int query, long long* info);
int pinn_launch_tile_hp64(int nd, int n2, const PinnKArgs* a, int grid, void* stream, int query, long long* info);
int pinn_launch_tile_hp128(int nd, int n2, const PinnKArgs* a, int grid, void* stream, int query, long long* info);
int pinn_launch_tile_hp256(int nd, int n2, const PinnKArgs* a, int grid, void* stream, int query, long long* info);
int pinn_launch_tile_hp512(int nd, int n2, const PinnKArgs* a, int grid, void* stream, int query, long long* info);

// streamed weight-gradient kernel of the WGX tile kernels (widths >= 128): same return codes; query fills info[0] (LDS bytes),
// info[1] (threads), info[3] (workgroups per CU)
int pinn_launch_wgrad_hp128(int nd, int n2, int comb, int mt, const PinnKArgs* a, int grid, void* stream, int query, long long* info);
int pinn_launch_wgrad_hp256(int nd, int n2, int comb, int mt, const PinnKArgs* a, int grid, void* stream, int query, long long* info);
int pinn_launch_wgrad_hp512(int nd, int n2, int comb, int mt, const PinnKArgs* a, int grid, void* stream, int query, long long* info);
