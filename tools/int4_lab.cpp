// In-graph timing of ao_int4_weight_int4pack_mm per Llama-3-8B shape (bs=1 by default) on cold weights:
// every launch in the captured graph uses a distinct weight set (> 512 MiB per shape, beyond the
// Infinity Cache), the graph is replayed and the wall time per launch reported.
//   int4_lab [-m M] [-g G] wpb:mode [wpb:mode ...]
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <string>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef int (*mm_fn)(const uint16_t*, const int32_t*, const uint16_t*, uint16_t*, int64_t, int64_t, int64_t, int, void*);
typedef int (*tune_fn)(int, int);
typedef const char* (*err_fn)(void);
typedef int (*trace_fn)(unsigned long long*);

__global__ void fill_kernel(uint32_t* p, size_t n, uint32_t seed, int kind) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    uint32_t h = (uint32_t)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    if (kind == 0) p[i] = h;                                                      // packed nibbles
    else if (kind == 1) p[i] = (0x3B00u + (h & 63)) | ((0x3A00u + ((h >> 8) & 63)) << 16);  // (scale, zero) bf16 pair
    else p[i] = (0x3F00u + (h & 0xff)) | (((h & 0x100) ? 0xBF00u : 0x3F00u) + ((h >> 9) & 0xff)) << 16;  // x ~ +-[0.5,1)
  }
}

int main(int argc, char** argv) {
  int M = 1, G = 128;
  setvbuf(stdout, nullptr, _IONBF, 0);
  std::vector<std::pair<int, int>> cfgs;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "-m")) M = atoi(argv[++i]);
    else if (!strcmp(argv[i], "-g")) G = atoi(argv[++i]);
    else { int w = 0, m = 0; sscanf(argv[i], "%d:%d", &w, &m); cfgs.push_back({w, m}); }
  }
  if (cfgs.empty()) cfgs.push_back({0, 0});
  std::string so = std::string(getenv("AO_LIB") ? getenv("AO_LIB") : "ao_amd/_C_mi355.so");
  void* h = dlopen(so.c_str(), RTLD_NOW);
  if (!h) { printf("dlopen failed: %s\n", dlerror()); return 1; }
  mm_fn mm = (mm_fn)dlsym(h, "ao_int4_weight_int4pack_mm");
  tune_fn tune = (tune_fn)dlsym(h, "ao_int4_set_tuning");
  err_fn lasterr = (err_fn)dlsym(h, "ao_last_error");
  trace_fn set_trace = (trace_fn)dlsym(h, "ao_int4_set_trace");
  struct Shape { const char* name; int64_t N, K; } shapes[] = {{"o", 4096, 4096}, {"qkv", 6144, 4096}, {"gate", 14336, 4096}, {"down", 4096, 14336}, {"gateup", 28672, 4096}};
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("M=%d G=%d   in-graph us/launch on cold weights (GB/s = algorithmic bytes / time)\n", M, G);
  for (auto& sh : shapes) {
    const size_t qbytes = sh.N * sh.K / 2, szbytes = (sh.K / G) * sh.N * 4;
    const int sets = (int)((640ull << 20) / qbytes) + 1;
    char *q, *sz; uint16_t *x, *y;
    CK(hipMalloc(&q, qbytes * sets)); CK(hipMalloc(&sz, szbytes * sets));
    CK(hipMalloc(&x, (size_t)M * sh.K * 2)); CK(hipMalloc(&y, (size_t)M * sh.N * 2 * sets));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, s, (uint32_t*)q, qbytes * sets / 4, 1u, 0);
    hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, s, (uint32_t*)sz, szbytes * sets / 4, 2u, 1);
    hipLaunchKernelGGL(fill_kernel, dim3(64), dim3(256), 0, s, (uint32_t*)x, (size_t)M * sh.K / 2, 3u, 2);
    CK(hipStreamSynchronize(s));
    const double abytes = (double)qbytes + szbytes + M * sh.K * 2.0 + M * sh.N * 2.0;
    std::vector<uint16_t> yref, ycur((size_t)M * sh.N);
    for (auto& c : cfgs) {
      tune(c.first, c.second);
      { int rc = mm(x, (const int32_t*)q, (const uint16_t*)sz, y, M, sh.N, sh.K, G, s);  // warm: lazy workspace alloc is not capturable
        if (rc) { printf("mm failed: %s\n", lasterr()); return 1; } CK(hipStreamSynchronize(s)); }
      if (c.second == 403) {
        // trace build: 100 MHz stamps; per workgroup [64]: wave 0 {entry, x landed, barrier, first blk, last blk, exit},
        // then from [8] per wave {first blk, last blk, exit}
        const int TS = 64, NWG = 1024;
        unsigned long long* tr; CK(hipMalloc(&tr, NWG * TS * 8)); CK(hipMemset(tr, 0, NWG * TS * 8));
        set_trace(tr);
        for (int i = 0; i < 3; ++i) mm(x, (const int32_t*)(q + qbytes * (i % sets)), (const uint16_t*)(sz + szbytes * (i % sets)), y, M, sh.N, sh.K, G, s);
        CK(hipStreamSynchronize(s));
        std::vector<unsigned long long> ht(NWG * TS); CK(hipMemcpy(ht.data(), tr, ht.size() * 8, hipMemcpyDeviceToHost));
        set_trace(nullptr); CK(hipFree(tr));
        unsigned long long t0 = ~0ull; int nwg = 0;
        for (int w = 0; w < NWG; ++w) if (ht[w * TS]) { nwg++; if (ht[w * TS] < t0) t0 = ht[w * TS]; }
        const char* names[6] = {"entry", "x landed", "barrier", "first blk", "last blk", "exit"};
        printf("%-5s trace, wave 0 of %d workgroups (us since first entry, min/mean/max): ", sh.name, nwg);
        for (int k = 0; k < 6; ++k) {
          double mn = 1e30, mx = 0, sum = 0;
          for (int w = 0; w < NWG; ++w) if (ht[w * TS]) { double v = (ht[w * TS + k] - t0) * 0.01; mn = v < mn ? v : mn; mx = v > mx ? v : mx; sum += v; }
          printf("%s %.2f/%.2f/%.2f  ", names[k], mn, sum / nwg, mx);
        }
        const char* wn[3] = {"first blk", "last blk", "exit"};
        printf("\n      all waves: ");
        for (int k = 0; k < 3; ++k) {
          double mn = 1e30, mx = 0, sum = 0; int n = 0;
          for (int w = 0; w < NWG; ++w) if (ht[w * TS]) for (int v = 0; v < 16; ++v) {
            unsigned long long t = ht[w * TS + 8 + 3 * v + k]; if (!t) continue;
            double u = (t - t0) * 0.01; mn = u < mn ? u : mn; mx = u > mx ? u : mx; sum += u; ++n; }
          printf("%s %.2f/%.2f/%.2f  ", wn[k], mn, n ? sum / n : 0.0, mx);
        }
        printf("\n");
        continue;
      }
      hipGraph_t g; hipGraphExec_t ge;
      CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
      for (int i = 0; i < sets; ++i) {
        int rc = mm(x, (const int32_t*)(q + qbytes * i), (const uint16_t*)(sz + szbytes * i), y + (size_t)M * sh.N * i, M, sh.N, sh.K, G, s);
        if (rc) { printf("mm failed: %s\n", lasterr()); return 1; }
      }
      CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, s));
      CK(hipStreamSynchronize(s));
      const int reps = 10;
      float best = 1e30f, tot = 0;
      for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); tot += ms; if (ms < best) best = ms;
      }
      const double us = tot / reps * 1e3 / sets, usb = best * 1e3 / sets;
      CK(hipMemcpy(ycur.data(), y + (size_t)M * sh.N * (sets - 1), ycur.size() * 2, hipMemcpyDeviceToHost));
      if (yref.empty()) yref = ycur;
      double num = 0, den = 0; int nbad = 0;
      for (size_t i = 0; i < ycur.size(); ++i) {
        uint32_t a = (uint32_t)ycur[i] << 16, b = (uint32_t)yref[i] << 16; float fa, fb; memcpy(&fa, &a, 4); memcpy(&fb, &b, 4);
        num += (double)(fa - fb) * (fa - fb); den += (double)fb * fb; nbad += ycur[i] != yref[i];
      }
      printf("%-5s N=%5ld K=%5ld wpb=%2d mode=%3d | %6.2f us (%5.0f GB/s)  best %6.2f us (%5.0f GB/s) | vs first cfg: rel %.2e, %d/%zu differ\n", sh.name, (long)sh.N, (long)sh.K,
             c.first, c.second, us, abytes / us / 1e3, usb, abytes / usb / 1e3, den > 0 ? sqrt(num / den) : 0.0, nbad, ycur.size());
      CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    CK(hipFree(q)); CK(hipFree(sz)); CK(hipFree(x)); CK(hipFree(y));
  }
  return 0;
}
