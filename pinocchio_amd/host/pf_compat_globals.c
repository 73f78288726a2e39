/* pf_compat_globals.c -- stand-alone definitions of the reference globals that
 * pf_compat.c uses (in the reference tree they come from src/variables.c). */
#include "pf_compat_types.h"

int ThisTask = 0, NTasks = 1;
product_data *products = 0;
static double *kdensity_slots[1] = {0};
double **kdensity = kdensity_slots;
smoothing_data Smoothing;
static grid_data grid0;
grid_data *MyGrids = &grid0;
ScaleDep_data ScaleDep;
cputime_data cputime;
param_data params;
double Rsmooth;
char date_string[25];
pf_spline_knots pf_invgrow_knots;
double (*pf_GrowingMode)(double, double) = 0;
double (*pf_GrowingMode_2LPT)(double, double) = 0;
double (*pf_GrowingMode_3LPT_1)(double, double) = 0;
double (*pf_GrowingMode_3LPT_2)(double, double) = 0;
int pf_compat_scale_dependent = 0;
pf_spline_knots pf_invgrow_knots_radius[64];
double **density = 0;
double ***first_derivatives = 0, ***second_derivatives = 0;
static pfft_complex *cvector_slots[1] = {0};
static double *rvector_slots[1] = {0};
pfft_complex **cvector_fft = cvector_slots;
double **rvector_fft = rvector_slots;
int pf_compat_lpt_order = 3;    /* stand-alone build: 3 = -DTWO_LPT -DTHREE_LPT, 2 = -DTWO_LPT, 1 = neither */
int pf_compat_tabulated_ct = 0;
int pf_compat_ct_interpolation = 0; /* 0 BILINEAR_SPLINE, 1 -DTRILINEAR, 2 -DALL_SPLINE */
int pf_compat_ell_sng = 0;
double (*pf_Hubble)(double) = 0;
double pf_compat_fr0 = 0.0;
