// Device -> pinned-host transfer on an SDMA engine (round 6).
// On this stack (ROCm 7.2, MI355X) hipMemcpyAsync of a device buffer into hipHostMalloc'ed memory runs as a SHADER BLIT
// (`__amd_rocclr_copyBuffer`, profiles/r06_d2h_blit_vs_sdma_probe.txt: every environment switch of the HIP / HSA runtimes tried leaves it a
// kernel): 105 MB of instance masks per image = a 1.9 ms kernel at PCIe rate whose few workgroups sit on CUs the whole time.  A GEMM launch with
// one 512-thread workgroup per CU (all of the tile kernels' launches) cannot place the workgroups of the occupied CUs until another CU
// drains -- a second round: measured +18 % on a GEMM loop with copies in flight, i.e. the copy's whole duration is ADDED to the GEMMs'.
// The HSA runtime underneath does own DMA engines; this entry asks it directly: hsa_amd_memory_async_copy between the agents that own the
// two allocations.  Blocking (returns when the bytes are in host memory) and NOT stream-ordered: the caller makes sure the source is
// complete (event / stream synchronize) and calls from a helper thread (ctypes releases the GIL); ape_amd/runtime.py does exactly that.
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <cstdint>
#include <cstdlib>
#include <vector>

void ape_set_error(const char* fmt, ...);

namespace {
bool owner_of(const void* p, hsa_agent_t* agent, bool* is_host) {
  hsa_amd_pointer_info_t info{};
  info.size = sizeof(info);
  if (hsa_amd_pointer_info(p, &info, nullptr, nullptr, nullptr) != HSA_STATUS_SUCCESS) return false;
  if (info.type != HSA_EXT_POINTER_TYPE_HSA && info.type != HSA_EXT_POINTER_TYPE_LOCKED) return false;
  hsa_device_type_t t = HSA_DEVICE_TYPE_CPU;
  if (hsa_agent_get_info(info.agentOwner, HSA_AGENT_INFO_DEVICE, &t) != HSA_STATUS_SUCCESS) return false;
  *agent = info.agentOwner;
  *is_host = t == HSA_DEVICE_TYPE_CPU;
  return true;
}
thread_local hsa_signal_t g_sig = {0};
}  // namespace

// 1 when both pointers belong to allocations the HSA runtime knows (device memory / pinned host memory) -- the precondition of the copy
extern "C" int ape_hip_sdma_usable(const void* host_dst, const void* dev_src) {
  hsa_agent_t a{}, b{};
  bool ha = false, hb = false;
  return (owner_of(host_dst, &a, &ha) && owner_of(dev_src, &b, &hb) && ha && !hb) ? 1 : 0;
}

// copy nbytes from device memory to pinned host memory on a DMA engine; returns when the bytes have landed (0 = ok)
extern "C" int ape_hip_sdma_d2h(void* host_dst, const void* dev_src, size_t nbytes) {
  if (nbytes == 0) return 0;
  hsa_agent_t dst_agent{}, src_agent{};
  bool dst_host = false, src_host = false;
  if (!owner_of(host_dst, &dst_agent, &dst_host) || !owner_of(dev_src, &src_agent, &src_host) || !dst_host || src_host) {
    ape_set_error("ape_hip_sdma_d2h: destination must be pinned host memory (hipHostMalloc) and source device memory of this process");
    return -1;
  }
  if (g_sig.handle == 0 && hsa_signal_create(1, 0, nullptr, &g_sig) != HSA_STATUS_SUCCESS) {
    g_sig.handle = 0;
    ape_set_error("ape_hip_sdma_d2h: hsa_signal_create failed");
    return -2;
  }
  hsa_signal_store_relaxed(g_sig, 1);
  const hsa_status_t rc = hsa_amd_memory_async_copy(host_dst, dst_agent, dev_src, src_agent, nbytes, 0, nullptr, g_sig);
  if (rc != HSA_STATUS_SUCCESS) {
    ape_set_error("ape_hip_sdma_d2h: hsa_amd_memory_async_copy failed (status 0x%x)", (unsigned)rc);
    return -2;
  }
  const hsa_signal_value_t v = hsa_signal_wait_scacquire(g_sig, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED);
  if (v < 0) {
    ape_set_error("ape_hip_sdma_d2h: the copy engine reported an error (signal %lld)", (long long)v);
    return -2;
  }
  return 0;
}

// Several device -> pinned-host copies in flight at once, each split into `parts` pieces: concurrent hsa_amd_memory_async_copy calls the
// runtime spreads over its copy engines (APE_SDMA_ENGINES=1: placed explicitly on the engines hsa_amd_memory_copy_engine_status reports
// free, through hsa_amd_memory_async_copy_on_engine).  One blocking copy after the other keeps ONE engine busy: 36-38 GB/s measured on the 1.18 GB per image of
// BASELINE's 1536^2 / top-500 configuration while kernels run (that configuration's step was bound by exactly this transfer:
// profiles/r06_config5_transfer.txt).  parts <= 0: APE_SDMA_PARTS or 2.  Returns when every byte has landed (0 = ok).
extern "C" int ape_hip_sdma_d2h_multi(int n, void* const* host_dst, const void* const* dev_src, const size_t* nbytes, int parts) {
  if (n <= 0) return 0;
  if (parts <= 0) {
    const char* e = getenv("APE_SDMA_PARTS");
    parts = e != nullptr ? atoi(e) : 2;
    if (parts <= 0) parts = 1;
  }
  if (parts > 8) parts = 8;
  struct Piece { void* dst; const void* src; size_t n; };
  std::vector<Piece> pieces;
  hsa_agent_t dst_agent{}, src_agent{};
  for (int i = 0; i < n; ++i) {
    if (nbytes[i] == 0) continue;
    bool dst_host = false, src_host = false;
    if (!owner_of(host_dst[i], &dst_agent, &dst_host) || !owner_of(dev_src[i], &src_agent, &src_host) || !dst_host || src_host) {
      ape_set_error("ape_hip_sdma_d2h_multi: copy %d: destination must be pinned host memory (hipHostMalloc) and source device memory of this process", i);
      return -1;
    }
    // pieces of at least 16 MB, boundaries on 4 KB
    int k = parts;
    while (k > 1 && nbytes[i] / k < ((size_t)16 << 20)) --k;
    const size_t step = ((nbytes[i] / k + 4095) / 4096) * 4096;
    for (size_t off = 0; off < nbytes[i]; off += step)
      pieces.push_back({(char*)host_dst[i] + off, (const char*)dev_src[i] + off, nbytes[i] - off < step ? nbytes[i] - off : step});
  }
  if (pieces.empty()) return 0;
  // engines free for device -> host
  std::vector<hsa_amd_sdma_engine_id_t> engines;
  if (parts > 1 || n > 1) {
    uint32_t mask = 0;
    if (hsa_amd_memory_copy_engine_status(dst_agent, src_agent, &mask) == HSA_STATUS_SUCCESS)
      for (uint32_t b = 1; b != 0 && b <= 0x8000u; b <<= 1)
        if (mask & b) engines.push_back((hsa_amd_sdma_engine_id_t)b);
    // APE_SDMA_ENGINES=1 (what ape_amd/runtime.py sets): the pieces go to DIFFERENT engines.  Unset / 0: plain concurrent
    // hsa_amd_memory_async_copy calls, which the runtime queues on ONE engine -- as fast as placed copies on an idle GPU (57 GB/s either
    // way, tools/gpu_sdma_multi_probe.py), but inside the running pipeline one engine moves ~37 GB/s and the 1536^2 / top-500 step stalls
    // exactly like the sequential form (profiles/r06_config5_transfer.txt: 29.9 vs 37.1 images/s).  The library's own default stays
    // "unplaced" because an explicit pick is a policy of the caller (on a multi-GPU node the caller may want to leave engines to xGMI).
    const char* e = getenv("APE_SDMA_ENGINES");
    if (e == nullptr || atoi(e) == 0) engines.clear();
  }
  static thread_local std::vector<hsa_signal_t> sigs;
  while (sigs.size() < pieces.size()) {
    hsa_signal_t sg{};
    if (hsa_signal_create(1, 0, nullptr, &sg) != HSA_STATUS_SUCCESS) {
      ape_set_error("ape_hip_sdma_d2h_multi: hsa_signal_create failed");
      return -2;
    }
    sigs.push_back(sg);
  }
  size_t issued = 0;
  int rcode = 0;
  for (size_t j = 0; j < pieces.size(); ++j) {
    hsa_signal_store_relaxed(sigs[j], 1);
    hsa_status_t rc = HSA_STATUS_ERROR;
    if (engines.size() >= 2)
      rc = hsa_amd_memory_async_copy_on_engine(pieces[j].dst, dst_agent, pieces[j].src, src_agent, pieces[j].n, 0, nullptr, sigs[j],
                                               engines[j % engines.size()], false);
    if (rc != HSA_STATUS_SUCCESS)
      rc = hsa_amd_memory_async_copy(pieces[j].dst, dst_agent, pieces[j].src, src_agent, pieces[j].n, 0, nullptr, sigs[j]);
    if (rc != HSA_STATUS_SUCCESS) {
      ape_set_error("ape_hip_sdma_d2h_multi: hsa_amd_memory_async_copy failed (status 0x%x)", (unsigned)rc);
      rcode = -2;
      break;
    }
    ++issued;
  }
  for (size_t j = 0; j < issued; ++j) {
    const hsa_signal_value_t v = hsa_signal_wait_scacquire(sigs[j], HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED);
    if (v < 0 && rcode == 0) {
      ape_set_error("ape_hip_sdma_d2h_multi: the copy engine reported an error (signal %lld)", (long long)v);
      rcode = -2;
    }
  }
  return rcode;
}

// number of copy engines the runtime reports free for device -> pinned host right now (measurement aid; -1: unknown)
extern "C" int ape_hip_sdma_engines(const void* host_dst, const void* dev_src) {
  hsa_agent_t a{}, b{};
  bool ha = false, hb = false;
  if (!owner_of(host_dst, &a, &ha) || !owner_of(dev_src, &b, &hb) || !ha || hb) return -1;
  uint32_t mask = 0;
  if (hsa_amd_memory_copy_engine_status(a, b, &mask) != HSA_STATUS_SUCCESS) return -1;
  return __builtin_popcount(mask);
}
