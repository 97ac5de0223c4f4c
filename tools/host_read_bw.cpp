#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#include <thread>
#include <cstdint>
#include <immintrin.h>
#include <atomic>
// pure streaming read of 48 x 8.3 MB maps with T threads, each map split into T contiguous slices
int main(int argc,char**argv){
  int T=argc>1?atoi(argv[1]):16; int M=48; size_t bytes=(size_t)1920*1080*4;
  std::vector<std::vector<char>> maps(M); for(auto&m:maps){m.resize(bytes); memset(m.data(),1,bytes);} 
  for(int rep=0;rep<3;++rep){
    auto t0=std::chrono::steady_clock::now();
    std::vector<std::thread> th; std::atomic<uint64_t> sink{0};
    for(int t=0;t<T;++t) th.emplace_back([&,t]{ uint64_t s=0; for(int m=0;m<M;++m){ size_t lo=bytes/T*t, hi=bytes/T*(t+1); const __m256i* p=(const __m256i*)(maps[m].data()+lo); __m256i acc=_mm256_setzero_si256(); for(size_t i=0;i<(hi-lo)/32;++i) acc=_mm256_add_epi64(acc,_mm256_loadu_si256(p+i)); s+=_mm256_extract_epi64(acc,0);} sink+=s;});
    for(auto&x:th)x.join();
    double dt=std::chrono::duration<double>(std::chrono::steady_clock::now()-t0).count();
    printf("T=%d pure read %.1f GB/s\n",T,M*bytes/dt/1e9);
  }
}
