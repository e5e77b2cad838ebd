// AddressSanitizer / UBSan harness for the host bitstream writer and the test-side stream parser (CPU only:
// GPU sanitizers are not available).  tools/sanitize/run.sh dumps a record with the oracle, builds this with
// -fsanitize=address,undefined and runs it: write, parse back, compare, then truncated and bit-flipped streams.
#include "../../include/wrenc_bitstream.h"
#include "../../oracle/vvc_parse.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
static std::vector<uint8_t> rd(const char* p){FILE*f=fopen(p,"rb");fseek(f,0,SEEK_END);long n=ftell(f);fseek(f,0,SEEK_SET);std::vector<uint8_t> v(n);if(fread(v.data(),1,n,f)!=(size_t)n)return{};fclose(f);return v;}
int main(int argc,char**argv){
  if(argc<4){fprintf(stderr,"usage: harness DIR WIDTH HEIGHT\n");return 2;}
  const std::string dir=argv[1];
  auto P=[&](const char*n){return dir+"/"+n;};
  auto a=rd(P("cu_log2_size.bin").c_str()),b=rd(P("luma_mode.bin").c_str()),c=rd(P("chroma_mode.bin").c_str()),d=rd(P("lev_y.bin").c_str()),e=rd(P("lev_cb.bin").c_str()),f=rd(P("lev_cr.bin").c_str());
  wrenc_bs_record r{a.data(),b.data(),c.data(),(const int16_t*)d.data(),(const int16_t*)e.data(),(const int16_t*)f.data()};
  const int W=atoi(argv[2]),H=atoi(argv[3]);
  std::vector<uint8_t> out(wrenc_bs_picture_bound(W,H)); size_t n=0,m=0;
  int rc=wrenc_bs_write_parameter_sets(W,H,32,out.data(),out.size(),&m);
  rc|=wrenc_bs_write_picture(W,H,32,7,&r,out.data()+m,out.size()-m,&n);
  printf("write rc %d bytes %zu\n",rc,n+m);
  std::vector<uint8_t> A(a.size()),B(b.size()),C(c.size()); std::vector<int16_t> D(d.size()/2),E(e.size()/2),F(f.size()/2);
  wro_picture_out po{nullptr,nullptr,nullptr,D.data(),E.data(),F.data(),A.data(),B.data(),C.data(),nullptr};
  int poc=0,qp=0; rc=wro_parse_picture(out.data(),n+m,0,&poc,&qp,&po);
  printf("parse rc %d poc %d qp %d same %d %d %d %d\n",rc,poc,qp,!memcmp(A.data(),a.data(),a.size()),!memcmp(B.data(),b.data(),b.size()),!memcmp(C.data(),c.data(),c.size()),!memcmp(D.data(),d.data(),d.size()));
  // truncated / damaged streams must fail cleanly
  for (size_t cut : {n+m-1, n+m-100, (n+m)/2, (size_t)150, (size_t)60, (size_t)7})
    if (cut < n+m) { rc=wro_parse_picture(out.data(),cut,0,&poc,&qp,&po); printf("cut %zu rc %d\n",cut,rc); }
  for (int k=0;k<40;k++){ std::vector<uint8_t> t(out.begin(),out.begin()+n+m); t[120+k*997%(n+m-200)]^=(uint8_t)(1+k); rc=wro_parse_picture(t.data(),t.size(),0,&poc,&qp,&po); if(k<5)printf("flip rc %d\n",rc);} 
  return 0; }
