// mpc_tile_module.hip — the MPC tile kernels as their own gfx950 code object (see mpc_tile_kernels.hip.h: the accumulator-register
// block needs the function attribute "amdgpu-agpr-alloc", which only a patched LLVM IR can carry).  Device code only; built by
// csrc/Makefile into ../mpc_tile.hsaco and embedded in libcrx.so / libcrx_x.so.  extern "C": the library finds the kernels by name.
#define CRX_MPC_TILE_MODULE 1
#include <hip/hip_runtime.h>
#include "mpc_tile_kernels.hip.h"

extern "C" __global__ void __launch_bounds__(64) crx_mpc_tile_kernel_s1(const crx::MpcTileArgs a) { crx::mpc_tile_body<24, 1>(a); }
extern "C" __global__ void __launch_bounds__(64) crx_mpc_tile_kernel_s2(const crx::MpcTileArgs a) { crx::mpc_tile_body<24, 2>(a); }
extern "C" __global__ void __launch_bounds__(64) crx_mpc_tile_refill_kernel_s1(const crx::MpcTileArgs a) { crx::mpc_tile_refill_body<24, 1>(a); }
extern "C" __global__ void __launch_bounds__(64) crx_mpc_tile_refill_kernel_s2(const crx::MpcTileArgs a) { crx::mpc_tile_refill_body<24, 2>(a); }
extern "C" __global__ void __launch_bounds__(64) crx_mpc_tile_phase_kernel_s1(const crx::MpcTileArgs a) { crx::mpc_tile_phase_body<24>(a); }
extern "C" __global__ void __launch_bounds__(64) crx_mpc_tile_lite_kernel(const crx::MpcTileArgs a) { crx::mpc_tile_body<24, 3>(a); }
