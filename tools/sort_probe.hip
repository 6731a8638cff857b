// Diagnostic: does rocprim::radix_sort_pairs with a partial bit range return a permutation?
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdio>
#include <vector>
#include <random>
#include <algorithm>
int main() {
  for (size_t n : {1000ul, 11200ul, 100000ul, 5000000ul})
    for (unsigned eb : {64u, 40u, 32u}) {
      const unsigned bb = 0;
      std::vector<uint64_t> k(n); std::vector<uint32_t> v(n);
      std::mt19937_64 g(n + bb);
      for (size_t i = 0; i < n; i++) { k[i] = eb == 64 ? g() : (g() >> (64 - eb)); v[i] = (uint32_t) i; }
      uint64_t *ki, *ko; uint32_t *vi, *vo;
      hipMalloc(&ki, n * 8); hipMalloc(&ko, n * 8); hipMalloc(&vi, n * 4); hipMalloc(&vo, n * 4);
      hipMemcpy(ki, k.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(vi, v.data(), n * 4, hipMemcpyHostToDevice);
      hipMemset(vo, 0xff, n * 4);
      size_t tb = 0; void* t = nullptr;
      rocprim::radix_sort_pairs(t, tb, (const uint64_t*) ki, ko, (const uint32_t*) vi, vo, n, bb, eb, 0);
      hipMalloc(&t, tb ? tb : 1);
      hipError_t e = rocprim::radix_sort_pairs(t, tb, (const uint64_t*) ki, ko, (const uint32_t*) vi, vo, n, bb, eb, 0);
      hipDeviceSynchronize();
      std::vector<uint64_t> ks(n); std::vector<uint32_t> vs(n);
      hipMemcpy(ks.data(), ko, n * 8, hipMemcpyDeviceToHost); hipMemcpy(vs.data(), vo, n * 4, hipMemcpyDeviceToHost);
      size_t bad_order = 0, bad_pair = 0;
      for (size_t i = 0; i < n; i++) {
        if (i && (ks[i] >> bb) < (ks[i - 1] >> bb)) bad_order++;
        if (vs[i] >= n || k[vs[i]] != ks[i]) bad_pair++;
      }
      std::vector<uint32_t> p = vs; std::sort(p.begin(), p.end());
      size_t dup = 0; for (size_t i = 0; i < n; i++) if (p[i] != i) dup++;
      printf("n=%zu end_bit=%u begin_bit=%u err=%d temp=%zu bad_order=%zu bad_pair=%zu not_perm=%zu\n", n, eb, bb, (int) e, tb, bad_order, bad_pair, dup);
      hipFree(ki); hipFree(ko); hipFree(vi); hipFree(vo); hipFree(t);
    }
  return 0;
}
