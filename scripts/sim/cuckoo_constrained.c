// CPU simulation behind DESIGN.md section 4 (arranged probe form): how often the kernel's random-walk cuckoo insertion cannot place a
// sketch when the second bucket shares `cb` low bits with the first.  usage: gcc -O2 -o sim cuckoo_constrained.c && ./sim <s> <tables> <cb>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
static uint64_t rs=88172645463325252ull;
static uint64_t rnd64(){ rs^=rs<<13; rs^=rs>>7; rs^=rs<<17; return rs*0x2545F4914F6CDD1Dull; }
static uint32_t b1(uint64_t x,uint32_t m){return (uint32_t)x&m;}
static uint32_t b2(uint64_t x,uint32_t m,int cb){uint32_t f=(uint32_t)(x>>20)&m; if(!cb) return f; uint32_t lm=(1u<<cb)-1; if(m<=lm*2) return f; uint32_t c=(f&~lm)|((uint32_t)x&lm); if(c==((uint32_t)x&m)) c^=(lm+1); return c;}
int main(int argc,char**argv){
  int s=atoi(argv[1]), trials=atoi(argv[2]), arr=atoi(argv[3]);
  uint32_t buckets=1; while(buckets<(uint32_t)s) buckets<<=1; uint32_t m=buckets-1;
  uint64_t *tab=malloc(16*buckets); int fails=0; long maxit=0; double totit=0;
  for(int t=0;t<trials;t++){
    for(uint32_t i=0;i<2*buckets;i++) tab[i]=~0ull;
    int failed=0;
    for(int e=0;e<s&&!failed;e++){
      uint64_t x=rnd64(); uint32_t b=b1(x,m); int placed=0; uint32_t r=(uint32_t)(x>>40)^(uint32_t)x;
      for(int it=0;it<4000;it++){
        if(tab[2*b]==~0ull){tab[2*b]=x;placed=1; if(it>maxit)maxit=it; totit+=it; break;}
        if(tab[2*b+1]==~0ull){tab[2*b+1]=x;placed=1; if(it>maxit)maxit=it; totit+=it; break;}
        r=r*1664525u+1013904223u; uint32_t sl=2*b+(r>>31); uint64_t y=tab[sl]; tab[sl]=x; x=y;
        uint32_t a=b1(x,m), c=b2(x,m,arr); b=(b==a)?c:a;
      }
      if(!placed) failed=1;
    }
    fails+=failed;
  }
  printf("s=%d arr=%d buckets=%u trials=%d fails=%d maxit=%ld avg_it=%.3f\n",s,arr,buckets,trials,fails,maxit,totit/((double)trials*s));
  return 0;
}
