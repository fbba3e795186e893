// microbenchmark: what bounds the all-edges relax kernels?  stream-only vs gather-only vs both, identity vs degree-sorted ids
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../include/vgl_hip.h"
#define CK(x) do{ if((x)!=0){ printf("err %s\n", vgl_hip_last_error()); exit(1);} }while(0)

template<int MODE, int MASK = -1>  // 0 stream adj+w only, 1 +gather dist[dst & MASK] (consecutive 8/thread), 2 gather with strided assignment
__global__ __launch_bounds__(256) void k(const int* adj, const float* w, const float* dist, long long E, float* out)
{
    float acc = 0;
    const long long e0 = (long long)blockIdx.x * 2048;
    if (MODE == 2) {
        for (int j = 0; j < 8; j++) { long long e = e0 + threadIdx.x + j*256; if (e < E) { int d = adj[e]; acc += w[e] + dist[d]; } }
    } else {
        const long long i0 = e0 + threadIdx.x * 8;
        if (i0 + 8 <= E) {
            const int4 a0 = *(const int4*)(adj + i0), a1 = *(const int4*)(adj + i0 + 4);
            const float4 w0 = *(const float4*)(w + i0), w1 = *(const float4*)(w + i0 + 4);
            acc = w0.x+w0.y+w0.z+w0.w+w1.x+w1.y+w1.z+w1.w;
            if (MODE == 1) acc += dist[a0.x&MASK]+dist[a0.y&MASK]+dist[a0.z&MASK]+dist[a0.w&MASK]+dist[a1.x&MASK]+dist[a1.y&MASK]+dist[a1.z&MASK]+dist[a1.w&MASK];
            else acc += (float)(a0.x^a0.y^a0.z^a0.w^a1.x^a1.y^a1.z^a1.w);
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}
int main(){
    vgl_hip_ctx* c; CK(vgl_hip_ctx_create(0, nullptr, &c));
    const int scale=24; const int V=1<<scale; const long long E=(long long)V*32;
    int *src,*dst,*s2,*d2,*adj,*fwd,*bwd; long long *rp; float *w,*dist,*out; long long* perm=nullptr;
    hipMalloc(&src,E*4); hipMalloc(&dst,E*4); hipMalloc(&s2,E*4); hipMalloc(&d2,E*4); hipMalloc(&adj,E*4); hipMalloc(&w,E*4);
    hipMalloc(&rp,(V+1)*8); hipMalloc(&fwd,V*4); hipMalloc(&bwd,V*4); hipMalloc(&dist,V*4); hipMalloc(&out,4);
    CK(vgl_hip_gen_rmat(c,scale,0,E,1,57,19,19,5,1,src,dst)); CK(vgl_hip_gen_weights(c,0,E,1,w));
    std::vector<float> h(V); for(int i=0;i<V;i++) h[i]=(float)(i%977); hipMemcpy(dist,h.data(),V*4,hipMemcpyHostToDevice);
    hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b);
    for (int ren=0; ren<2; ren++){
        long long kept;
        if (ren) { CK(vgl_hip_degree_order(c,V,E,src,dst,2,fwd,bwd)); CK(vgl_hip_relabel_i32(c,E,fwd,src,s2)); CK(vgl_hip_relabel_i32(c,E,fwd,dst,d2)); CK(vgl_hip_coo_to_csr(c,V,E,s2,d2,0,V,(int64_t*)rp,adj,nullptr,(int64_t*)&kept)); }
        else CK(vgl_hip_coo_to_csr(c,V,E,src,dst,0,V,(int64_t*)rp,adj,nullptr,(int64_t*)&kept));
        const unsigned nt=(unsigned)((E+2047)/2048);
        for (int m=0;m<4;m++){ float best=1e9; for(int r=0;r<5;r++){ hipEventRecord(a);
            if(m==0) k<1,0xFFFF><<<nt,256>>>(adj,w,dist,E,out); else if(m==1) k<1,0xFFFFF><<<nt,256>>>(adj,w,dist,E,out); else if (m==2) k<1,0x3FFFFF><<<nt,256>>>(adj,w,dist,E,out); else k<1,0xFF><<<nt,256>>>(adj,w,dist,E,out);
            hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms,a,b); if(ms<best)best=ms; }
            const char* names[4]={"64K entries (256 KB)","1M entries (4 MB)","4M entries (16 MB)","256 entries (1 KB)"};
            printf("renumber=%d gather from %s: %.3f ms\n", ren, names[m], best); }
        for (int mode=0; mode<3; mode++){
            float best=1e9;
            for(int r=0;r<5;r++){ hipEventRecord(a);
                if(mode==0) k<0><<<nt,256>>>(adj,w,dist,E,out); else if(mode==1) k<1><<<nt,256>>>(adj,w,dist,E,out); else k<2><<<nt,256>>>(adj,w,dist,E,out);
                hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms,a,b); if(ms<best)best=ms; }
            printf("renumber=%d mode=%d  %.3f ms  (%.1f GB/s stream-equivalent)\n", ren, mode, best, E*8.0/best/1e6);
        }
    }
    return 0;
}
