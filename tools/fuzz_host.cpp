// tools/fuzz_host.cpp — robustness of the host entropy front-end: mutated .ogg files (bit flips, byte overwrites, truncation, byte copies; page CRCs
// re-computed for 3 of 4 so that the damage reaches the codec layer) through decode_corpus(entropy_only): every file must end in ok or a clean error.
// Build with sanitizers (CPU only):  g++ -std=c++17 -O1 -g -fsanitize=address,undefined tools/fuzz_host.cpp parseoggvorbis_amd/host/{ParseOggVorbis,CorpusDecoder,hooks}.cpp \
//   -Lparseoggvorbis_amd/csrc -lvorbis_synth_hip -Wl,-rpath,$PWD/parseoggvorbis_amd/csrc -Wl,-rpath-link,/opt/rocm/lib -lpthread -o /tmp/fuzz_host
// Run: /tmp/fuzz_host <iterations of 16 files> <seed> file.ogg ...     (round 1: 14 400 mutated files, no finding)
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <random>
#include <vector>
#include "../parseoggvorbis_amd/host/CorpusDecoder.hpp"
static std::vector<uint8_t> rd(const char* p){FILE*f=fopen(p,"rb");fseek(f,0,SEEK_END);long n=ftell(f);fseek(f,0,SEEK_SET);std::vector<uint8_t> b(n);fread(b.data(),1,n,f);fclose(f);return b;}
// recompute page CRCs so that mutations reach the codec layer
static uint32_t crc_tab[256];
static void crc_init(){for(uint32_t i=0;i<256;i++){uint32_t r=i<<24;for(int j=0;j<8;j++)r=(r&0x80000000u)?(r<<1)^0x04c11db7u:(r<<1);crc_tab[i]=r;}}
static void fix_crcs(std::vector<uint8_t>& b){size_t o=0;while(o+27<=b.size()&&memcmp(&b[o],"OggS",4)==0){uint32_t ns=b[o+26];if(o+27+ns>b.size())break;size_t len=27+ns;for(uint32_t i=0;i<ns;i++)len+=b[o+27+i];if(o+len>b.size())break;b[o+22]=b[o+23]=b[o+24]=b[o+25]=0;uint32_t c=0;for(size_t i=0;i<len;i++)c=(c<<8)^crc_tab[((c>>24)&0xff)^b[o+i]];b[o+22]=c;b[o+23]=c>>8;b[o+24]=c>>16;b[o+25]=c>>24;o+=len;}}
int main(int argc,char**argv){
  crc_init();
  int iters=atoi(argv[1]); std::mt19937 rng(atoi(argv[2]));
  std::vector<std::vector<uint8_t>> base; for(int i=3;i<argc;i++) base.push_back(rd(argv[i]));
  size_t ok=0,err=0;
  for(int it=0;it<iters;++it){
    std::vector<std::vector<uint8_t>> muts; std::vector<CorpusItem> items;
    for(int k=0;k<16;k++){ auto b=base[rng()%base.size()]; int nm=1+rng()%8; bool fix=rng()%4!=0;
      for(int m=0;m<nm;m++){ size_t pos=rng()%b.size(); int kind=rng()%4; if(kind==0)b[pos]^=1u<<(rng()%8); else if(kind==1)b[pos]=rng(); else if(kind==2&&b.size()>100)b.resize(b.size()-rng()%(b.size()/2)); else { size_t a=rng()%b.size(); b[pos]=b[a]; } }
      if(fix) fix_crcs(b); muts.push_back(std::move(b)); }
    for(auto&m:muts) items.push_back(CorpusItem{m.data(),m.size()});
    CorpusOptions o; o.threads=2; o.entropy_only=true; std::vector<CorpusFileResult> res; CorpusStats st;
    OkOrError r=decode_corpus(items,o,nullptr,res,&st);
    if(r.is_error_){printf("run error %s\n",r.err_msg_.c_str());return 1;}
    for(auto&x:res){ if(x.status.is_error_)err++; else ok++; }
  }
  printf("ok=%zu err=%zu\n",ok,err); return 0; }
