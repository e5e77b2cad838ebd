// ASan / UBSan harness for the CPU oracle itself: search + final pass + decoder-side reconstruction of small
// pictures at every depth (built and run by tools/sanitize/run.sh).
#include "../../oracle/wrenc_oracle.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
int main(){
  const int W=96,H=64;
  for (int depth=0; depth<4; ++depth) for (int qp : {22, 37}) {
    std::vector<uint8_t> y(W*H), cb(W*H/4), cr(W*H/4);
    srand(depth*100+qp);
    for (int i=0;i<W*H;i++) y[i]=(uint8_t)(128+60*((i%W)/7%2)+rand()%40-20);
    for (int i=0;i<W*H/4;i++){cb[i]=(uint8_t)(y[(i/(W/2))*2*W+(i%(W/2))*2]/2+40); cr[i]=(uint8_t)(200-cb[i]/2+rand()%5);}
    std::vector<uint8_t> ry(W*H),rcb(W*H/4),rcr(W*H/4),sz(W*H/16),lm(W*H/16),cm(W*H/64); std::vector<int16_t> ly(W*H),lcb(W*H/4),lcr(W*H/4); std::vector<float> cost(W*H/1024);
    wro_picture_out o{ry.data(),rcb.data(),rcr.data(),ly.data(),lcb.data(),lcr.data(),sz.data(),lm.data(),cm.data(),cost.data()};
    wro_params p{W,H,qp,depth};
    int rc=wro_encode_picture(&p,y.data(),cb.data(),cr.data(),&o);
    std::vector<uint8_t> dy(W*H),dcb(W*H/4),dcr(W*H/4);
    int rc2=wro_reconstruct_from_record(&p,&o,dy.data(),dcb.data(),dcr.data());
    printf("depth %d qp %d rc %d %d recon same %d mism %ld\n",depth,qp,rc,rc2,dy==ry&&dcb==rcb&&dcr==rcr,wro_last_final_pass_mismatches());
  }
}
