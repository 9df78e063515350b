// ipc_probe.cpp -- what does hipIpc offer between two processes that share ONE device on this stack?
//   hipcc -O2 tools/ipc_probe.cpp -o tools/ipc_probe.bin;  ./ipc_probe.bin server /tmp/h & ./ipc_probe.bin client /tmp/h
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unistd.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); printf("  %-70s -> %s\n", #x, hipGetErrorString(e_)); fflush(stdout); (void) hipGetLastError(); } while (0)
struct Msg { hipIpcMemHandle_t mem; hipIpcEventHandle_t ev; size_t off, bytes; };
int main(int argc, char** argv) {
  const std::string mode = argv[1], path = argv[2];
  const size_t n = argc > 3 ? atol(argv[3]) : (1 << 20), off = argc > 4 ? atol(argv[4]) : 4096 * 3 + 64;
  if (mode == "server") {
    char* d = nullptr;
    CK(hipMalloc((void**) &d, n));
    CK(hipMemset(d, 0x5a, n));
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming | hipEventInterprocess));
    Msg m; memset(&m, 0, sizeof m);
    hipDeviceptr_t base; size_t size;
    CK(hipMemGetAddressRange(&base, &size, d + off));
    printf("server: base %p size %zu (d %p)\n", base, size, (void*) d);
    CK(hipIpcGetMemHandle(&m.mem, base));
    CK(hipIpcGetEventHandle(&m.ev, ev));
    CK(hipEventRecord(ev, nullptr));
    m.off = off; m.bytes = n - off < 65536 ? n - off : 65536;
    FILE* f = fopen((path + ".tmp").c_str(), "wb"); fwrite(&m, sizeof m, 1, f); fclose(f);
    rename((path + ".tmp").c_str(), path.c_str());
    while (access((path + ".done").c_str(), F_OK) != 0) usleep(10000);
    CK(hipFree(d));
    printf("server done\n");
  } else {
    while (access(path.c_str(), F_OK) != 0) usleep(10000);
    Msg m; FILE* f = fopen(path.c_str(), "rb"); fread(&m, sizeof m, 1, f); fclose(f);
    void* p = nullptr; hipEvent_t ev = nullptr;
    CK(hipIpcOpenMemHandle(&p, m.mem, hipIpcMemLazyEnablePeerAccess));
    CK(hipIpcOpenEventHandle(&ev, m.ev));
    printf("client: mapped %p\n", p);
    hipPointerAttribute_t at; memset(&at, 0, sizeof at);
    CK(hipPointerGetAttributes(&at, p));
    printf("client: attr type %d device %d devptr %p\n", (int) at.type, at.device, at.devicePointer);
    hipDeviceptr_t base; size_t size = 0;
    CK(hipMemGetAddressRange(&base, &size, p));
    printf("client: range base %p size %zu\n", base, size);
    char* mine = nullptr;
    CK(hipMalloc((void**) &mine, m.bytes));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipStreamWaitEvent(s, ev, 0));
    CK(hipMemcpyAsync(mine, (char*) p + m.off, m.bytes, hipMemcpyDeviceToDevice, s));
    CK(hipMemcpyAsync(mine, (char*) p + m.off, m.bytes, hipMemcpyDefault, s));
    CK(hipMemcpyAsync(mine, (char*) p, m.bytes, hipMemcpyDeviceToDevice, s));
    CK(hipMemcpyDtoDAsync(mine, (char*) p + m.off, m.bytes, s));
    CK(hipMemcpyPeerAsync(mine, 0, (char*) p + m.off, 0, m.bytes, s));
    CK(hipStreamSynchronize(s));
    std::vector<char> h(m.bytes);
    CK(hipMemcpy(h.data(), mine, m.bytes, hipMemcpyDeviceToHost));
    printf("client: first bytes %02x %02x\n", (unsigned char) h[0], (unsigned char) h[m.bytes - 1]);
    hipEvent_t done; CK(hipEventCreateWithFlags(&done, hipEventDisableTiming | hipEventInterprocess));
    CK(hipEventRecord(done, s));
    CK(hipIpcCloseMemHandle(p));
    f = fopen((path + ".done").c_str(), "wb"); fclose(f);
    printf("client done\n");
  }
  return 0;
}
