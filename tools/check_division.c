#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <stdlib.h>
#include <omp.h>
static inline float u2f(uint32_t u){float f;memcpy(&f,&u,4);return f;}
static inline uint32_t f2u(float f){uint32_t u;memcpy(&u,&f,4);return u;}
static inline uint64_t rng(uint64_t *s){uint64_t x=*s;x^=x<<13;x^=x>>7;x^=x<<17;return *s=x;}
int main(int argc,char**argv){
  long long bad=0, total=0;
  // part 1: for a set of divisors, ALL numerators with exponent in [2^-44, 2^16) both signs + zero
  int nd = argc>1?atoi(argv[1]):64;
  uint64_t seed=88172645463325252ull;
  float *ds = malloc(sizeof(float)*nd);
  for(int i=0;i<nd;i++){
    // divisors 2R in [1, 128]: random mantissas, plus mantissa all ones / all zeros / protein radii
    uint32_t m = (uint32_t)rng(&seed)&0x7FFFFF; int e = 127 + (int)(rng(&seed)%8);
    if(i==0) m=0x7FFFFF; if(i==1) m=0; if(i==2) m=0x7FFFFE; if(i==3) m=1;
    ds[i]=u2f((uint32_t)e<<23|m);
    if(i==4) ds[i]=2.0f*(1.88f+1.4f); if(i==5) ds[i]=2.0f*(1.61f+1.4f); if(i==6) ds[i]=2.0f*(1.42f+1.4f); if(i==7) ds[i]=2.0f*(1.64f+1.4f);
    if(i==8) ds[i]=2.0f*(1.76f+1.4f); if(i==9) ds[i]=2.0f*(1.46f+1.4f); if(i==10) ds[i]=2.0f*(1.77f+1.4f); if(i==11) ds[i]=2.0f*(0.5f);
  }
  for(int i=0;i<nd;i++){
    float d=ds[i]; volatile float one=1.0f; float y=one/d;
    long long b=0;
    #pragma omp parallel for reduction(+:b) schedule(static)
    for(long long k=0;k<(1ll<<32);k++){
      uint32_t u=(uint32_t)k; uint32_t ex=(u>>23)&0xFF;
      if(ex!=0 && (ex<127-44 || ex>=127+16)) continue;
      if(ex==0 && (u&0x7FFFFF)) continue; // denormal numerators cannot occur (see DESIGN)
      float a=u2f(u);
      float q0=a*y; float r=fmaf(-q0,d,a); float q1=fmaf(r,y,q0);
      float ref=a/d;
      if(f2u(q1)!=f2u(ref)) { b++; if(b<3) printf("MISMATCH a=%a d=%a q1=%a ref=%a\n",a,d,q1,ref);}
    }
    bad+=b; total+=(1ll<<32);
    if(b||i<12) printf("d=%a (%g): mismatches %lld\n",d,d,b);
  }
  printf("part1 divisors %d mismatches %lld\n",nd,bad);
  return bad!=0;
}
