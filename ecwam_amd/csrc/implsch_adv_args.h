// Arguments of the one-kernel WAMINTGR step (the advecting tile load of k_implsch4, implsch_v4.h::V4Adv) in the untyped form capi.hip hands
// to the launcher of implsch4a.hip.
#pragma once
struct Implsch4AdvArgs {
  const void* f_in;                     // rows the stencil reads
  const int *klon, *klat, *kcor;        // neighbour tables of the owned points
  const void* cg;                       // CGROUP_EXT
  const void* pt;                       // per-point scalars of the weights (k_ctu_prep)
  const void* dirT;                     // per-direction factors (k_ctu_prep)
  const int* dirI;
  double xdella, delpro;
  const void* gin;                      // fast waves after their sub-steps: compact rows [rows][NANG][gin_k] (or NULL)
  double delpro_lf;
  int gin_k, mlf;                       // frequencies [0, mlf) advance with delpro_lf
  const void* obs;                      // LSUBGRID: OBS[n][8][NFRE] (or NULL)
  void* gfast;                          // compact rows that also receive the first gfast_k frequencies of the new spectrum (or NULL)
  int gfast_k;
  int m0, m1;                           // advected frequencies [m0, m1)
  int xcd_walk;
  int mode;                             // 1 = the product; 2 = the go / no-go probe (made-up weights; builds with -DV4_ADV_PROBE only)
};
