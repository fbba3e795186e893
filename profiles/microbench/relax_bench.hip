// where does the time of an all-edges relax pass go?  variants of the tile kernel on RMAT-24 (degree-sorted ids)
#include "../vectorgraphlibrary_amd/csrc/vgl_hip_internal.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cfloat>
#define CK(x) do{ if((x)!=0){ printf("err %s\n", vgl_hip_last_error()); exit(1);} }while(0)
int vgl_set_error(const char*, int, const char*) { return 1; }

template<bool ROWMAP, bool SRC, bool GATHER>
__global__ __launch_bounds__(VGL_BLOCK) void k(const int64_t* rowptr, const int* adj, const float* w, const int32_t* tile_row, int64_t E, float* dist, float* out)
{
    __shared__ int s_map[VGL_TILE];
    __shared__ int s_w[VGL_WAVES];
    const int64_t e0 = (int64_t)blockIdx.x * VGL_TILE;
    const int n = (int)min((int64_t)VGL_TILE, E - e0);
    const int r_first = tile_row[blockIdx.x], r_last = tile_row[blockIdx.x + 1];
    if (ROWMAP) vgl_tile_row_map(s_map, s_w, rowptr, e0, r_first, r_last);
    const int i0 = threadIdx.x * VGL_EPT;
    float acc = 0;
    if (i0 + VGL_EPT <= n) {
        const int4 a0 = *(const int4*)(adj + e0 + i0), a1 = *(const int4*)(adj + e0 + i0 + 4);
        const float4 w0 = *(const float4*)(w + e0 + i0), w1 = *(const float4*)(w + e0 + i0 + 4);
        int dsts[8] = {a0.x,a0.y,a0.z,a0.w,a1.x,a1.y,a1.z,a1.w};
        float ws[8] = {w0.x,w0.y,w0.z,w0.w,w1.x,w1.y,w1.z,w1.w};
        float d = 1.0f; int prev = -1;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (ROWMAP && SRC) { int row = s_map[i0 + j]; if (row != prev) { prev = row; d = dist[r_first + row]; } }
            float o = GATHER ? dist[dsts[j]] : (float)dsts[j];
            acc += (o > d + ws[j]) ? 1.0f : 0.0f;
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}
int main(){
    vgl_hip_ctx* c; CK(vgl_hip_ctx_create(0, nullptr, &c));
    const int scale=24; const int V=1<<scale; const long long E=(long long)V*32;
    int *src,*dst,*s2,*d2,*adj,*fwd,*bwd; int64_t *rp; float *w,*dist,*out;
    hipMalloc(&src,E*4); hipMalloc(&dst,E*4); hipMalloc(&s2,E*4); hipMalloc(&d2,E*4); hipMalloc(&adj,E*4); hipMalloc(&w,E*4);
    hipMalloc(&rp,(V+1)*8); hipMalloc(&fwd,V*4); hipMalloc(&bwd,V*4); hipMalloc(&dist,V*4); hipMalloc(&out,4);
    CK(vgl_hip_gen_rmat(c,scale,0,E,1,57,19,19,5,1,src,dst)); CK(vgl_hip_gen_weights(c,0,E,1,w));
    std::vector<float> h(V); for(int i=0;i<V;i++) h[i]=(float)(i%977); hipMemcpy(dist,h.data(),V*4,hipMemcpyHostToDevice);
    int64_t kept;
    CK(vgl_hip_degree_order(c,V,E,src,dst,2,fwd,bwd)); CK(vgl_hip_relabel_i32(c,E,fwd,src,s2)); CK(vgl_hip_relabel_i32(c,E,fwd,dst,d2));
    CK(vgl_hip_coo_to_csr(c,V,E,s2,d2,0,V,rp,adj,nullptr,&kept));
    vgl_hip_graph* g; CK(vgl_hip_graph_create(c,V,0,V,rp,adj,E,nullptr,nullptr,0,&g));
    const int32_t* tile_row; int64_t ntiles; CK(vgl_hip_graph_tile_rows(g,0,&tile_row,&ntiles));
    hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b);
    auto run=[&](const char* name, auto kern){ float best=1e9; for(int r=0;r<5;r++){ hipEventRecord(a); kern(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms,a,b); if(ms<best)best=ms;} printf("%-32s %.3f ms\n", name, best); };
    unsigned nt=(unsigned)ntiles;
    run("stream only", [&]{ k<false,false,false><<<nt,256>>>(rp,adj,w,tile_row,E,dist,out); });
    run("stream + rowmap", [&]{ k<true,false,false><<<nt,256>>>(rp,adj,w,tile_row,E,dist,out); });
    run("stream + rowmap + dist[src]", [&]{ k<true,true,false><<<nt,256>>>(rp,adj,w,tile_row,E,dist,out); });
    run("stream + gather (no rowmap)", [&]{ k<false,false,true><<<nt,256>>>(rp,adj,w,tile_row,E,dist,out); });
    run("stream + rowmap + src + gather", [&]{ k<true,true,true><<<nt,256>>>(rp,adj,w,tile_row,E,dist,out); });
    return 0;
}
