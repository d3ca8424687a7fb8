// the pieces of the host emulation build that are not in pl-svo_amd/csrc/: the dynamic-LDS arrays the kernels declare
// `extern __shared__` (one definition per name; every launch poisons the bytes it asked for), and a marker symbol by which anything
// can tell this library from the product (bench.py refuses it)
#include <hip/hip_runtime.h>
namespace plsvo_hip {
thread_local __attribute__((aligned(16))) unsigned char smem[wave_emu::DYNAMIC_LDS_BYTES];              // align_kernels.hip
thread_local __attribute__((aligned(16))) int s_winner[wave_emu::DYNAMIC_LDS_BYTES / sizeof(int)];      // chain_kernels.hip
}
void wave_emu::poison_dynamic_lds(size_t bytes) {
  if (bytes > DYNAMIC_LDS_BYTES) bytes = DYNAMIC_LDS_BYTES;
  memset(plsvo_hip::smem, 0xCD, bytes);
  memset(plsvo_hip::s_winner, 0xCD, bytes);
}
extern "C" int plsvo_emu_build(void) { return 1; }
