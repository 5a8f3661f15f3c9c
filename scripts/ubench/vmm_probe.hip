// Probe (not product code): does this ROCm stack support the virtual-memory-management API across PROCESSES -- one contiguous virtual
// range whose pieces are physical allocations of different processes (hipMemCreate / hipMemExportToShareableHandle (POSIX fd) /
// hipMemImportFromShareableHandle / hipMemMap / hipMemSetAccess)?  That is what a row-sharded embedding table needs to look like ONE
// table to an unchanged gather kernel (BASELINE config 4: "27 M-row table, row-sharded across 8 x MI355X"): loads of a peer's rows go
// over xGMI, no collective.  Two processes sharing the one device stand in for two GPUs.
//   build: hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/vmm_probe scripts/ubench/vmm_probe.hip
#include <hip/hip_runtime.h>
#include <sys/socket.h>
#include <sys/wait.h>
#include <unistd.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("FAIL %s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_fill(float* p, size_t n, float v) { for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) p[i] = v + (float)(i & 1023); }
__global__ void k_sum(const float* p, size_t n, double* out) {
    double s = 0;
    for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) s += p[i];
    atomicAdd(out, s);
}
static int send_fd(int sock, int fd) {
    char buf[1] = {0}; struct iovec io = {buf, 1}; char c[CMSG_SPACE(sizeof(int))]; memset(c, 0, sizeof(c));
    struct msghdr m = {}; m.msg_iov = &io; m.msg_iovlen = 1; m.msg_control = c; m.msg_controllen = sizeof(c);
    struct cmsghdr* h = CMSG_FIRSTHDR(&m); h->cmsg_level = SOL_SOCKET; h->cmsg_type = SCM_RIGHTS; h->cmsg_len = CMSG_LEN(sizeof(int));
    memcpy(CMSG_DATA(h), &fd, sizeof(int));
    return sendmsg(sock, &m, 0) == 1 ? 0 : -1;
}
static int recv_fd(int sock) {
    char buf[1]; struct iovec io = {buf, 1}; char c[CMSG_SPACE(sizeof(int))];
    struct msghdr m = {}; m.msg_iov = &io; m.msg_iovlen = 1; m.msg_control = c; m.msg_controllen = sizeof(c);
    if (recvmsg(sock, &m, 0) != 1) return -1;
    struct cmsghdr* h = CMSG_FIRSTHDR(&m); int fd = -1; if (h) memcpy(&fd, CMSG_DATA(h), sizeof(int));
    return fd;
}
static int role(int rank, int sock) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    prop.requestedHandleType = hipMemHandleTypePosixFileDescriptor;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    const size_t shard = ((size_t)64 << 20) / gran * gran;                     // 64 MB per rank
    printf("rank %d: granularity %zu bytes, shard %zu\n", rank, gran, shard);
    void* base = nullptr;
    CK(hipMemAddressReserve(&base, 2 * shard, 0, nullptr, 0));
    hipMemGenericAllocationHandle_t mine, theirs;
    CK(hipMemCreate(&mine, shard, &prop, 0));
    CK(hipMemMap((char*)base + rank * shard, shard, 0, mine, 0));
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess((char*)base + rank * shard, shard, &acc, 1));
    hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, 0, (float*)((char*)base + rank * shard), shard / 4, (float)(1000 * (rank + 1)));
    CK(hipDeviceSynchronize());
    int fd = -1;
    CK(hipMemExportToShareableHandle(&fd, mine, hipMemHandleTypePosixFileDescriptor, 0));
    if (rank == 0) { if (send_fd(sock, fd)) { printf("send_fd failed\n"); return 1; } }
    int pfd = recv_fd(sock);
    if (rank == 1) { if (send_fd(sock, fd)) { printf("send_fd failed\n"); return 1; } }
    if (pfd < 0) { printf("rank %d: no fd received\n", rank); return 1; }
    int ver = 0; (void)hipRuntimeGetVersion(&ver);
    printf("rank %d: HIP runtime %d\n", rank, ver);
    // (ROCm 7.2: the descriptor's VALUE cast to a pointer, as in CUDA; the 7.0 runtime bundled with PyTorch 2.10 wants a pointer TO it)
    if (ver >= 70200000) CK(hipMemImportFromShareableHandle(&theirs, (void*)(uintptr_t)pfd, hipMemHandleTypePosixFileDescriptor));
    else CK(hipMemImportFromShareableHandle(&theirs, &pfd, hipMemHandleTypePosixFileDescriptor));
    CK(hipMemMap((char*)base + (1 - rank) * shard, shard, 0, theirs, 0));
    CK(hipMemSetAccess((char*)base + (1 - rank) * shard, shard, &acc, 1));
    double* out; CK(hipMalloc(&out, 8)); CK(hipMemset(out, 0, 8));
    hipLaunchKernelGGL(k_sum, dim3(1024), dim3(256), 0, 0, (const float*)base, 2 * shard / 4, out);     // ONE kernel over both shards
    double s = 0; CK(hipMemcpy(&s, out, 8, hipMemcpyDeviceToHost));
    const double n = (double)(shard / 4), want = n * 1000 + n * 2000 + 2 * n * 511.5;
    printf("rank %d: sum over the contiguous range %.0f, expected %.0f -> %s\n", rank, s, want, s == want ? "OK" : "MISMATCH");
    char b = 0; (void)!write(sock, &b, 1); (void)!read(sock, &b, 1);           // nobody unmaps before the peer has read
    return s == want ? 0 : 1;
}
int main() {
    int sv[2];
    if (socketpair(AF_UNIX, SOCK_STREAM, 0, sv)) { perror("socketpair"); return 1; }
    pid_t pid = fork();                                                         // before any HIP call: each process owns its runtime
    if (pid == 0) { close(sv[0]); int rc = role(1, sv[1]); fflush(stdout); _exit(rc); }
    close(sv[1]);
    int rc = role(0, sv[0]);
    int st = 0; waitpid(pid, &st, 0);
    printf("VMM across processes: %s\n", (rc == 0 && WIFEXITED(st) && WEXITSTATUS(st) == 0) ? "WORKS" : "DOES NOT WORK HERE");
    return rc;
}
