// saigehip.hip -- MI355X (gfx950) kernels + C ABI of the SAIGEgds single-variant
// scan.  Interface and reference citations: include/saigehip.h; design,
// data layout and rooflines: DESIGN.md.
//
// Math of the score stage.  For a variant with (imputed, flipped) dosage G the
// reference computes (src/saige_main.cpp:300-350, either branch)
//     adj = G - X c',  c' = (X'VX)^-1 X'V G,  S = sum y_mu*adj,
//     var = r * sum mu2*adj^2.
// Both branches reduce to four carrier-only sums over I = {i : G_i != 0}:
//     c'_k = sum_I G_i A[i,k]          A = t_XVX_inv_XV
//     e_k  = sum_I G_i mu2_i X[i,k]
//     s    = sum_I G_i y_mu_i
//     w    = sum_I G_i^2 mu2_i
//     var2 = c'^T XVX c' + w - 2 e.c'        S = s - S_a.c'
// (quantitative: mu2 == 1).  The per-sample vector F[i] = [A[i,:], mu2_i X[i,:],
// y_mu_i, mu2_i] (P = 2K+2 doubles, one 16-byte aligned row) is what a carrier
// costs in memory traffic.

#include <hip/hip_runtime.h>
#include <math.h>
#include <cmath>
#include <algorithm>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "saigehip.h"

#define KMAX SGX_MAX_COEFF
#define WAVE 64

// ---------------------------------------------------------------------------
// error handling

static thread_local std::string g_err;

static int fail(int code, const char *fmt, ...)
{
	char buf[512];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof buf, fmt, ap);
	va_end(ap);
	g_err = buf;
	return code;
}

#define HIPCHK(expr)                                                              \
	do {                                                                          \
		hipError_t e_ = (expr);                                                   \
		if (e_ != hipSuccess)                                                     \
			return fail(SGX_EHIP, "%s failed: %s (%s:%d)", #expr,                 \
				hipGetErrorString(e_), __FILE__, __LINE__);                       \
	} while (0)

#include "dev_common.h"
#include "kern_score.h"
#include "kern_score_mfma.h"
#include "kern_score3.h"
#include "kern_spa.h"
#include "kern_spa2.h"
#include "kern_spa4.h"
#include "kern_synth.h"
#include "kern_grm.h"
#include "kern_burden.h"
#include "kern_pack.h"

// ---------------------------------------------------------------------------
// host side: one translation unit (every kernel template is instantiated once), in topic files
#include "host_state.h"
#include "host_init.h"
#include "host_spa.h"
#include "host_lanes.h"
#include "host_blocks.h"
#include "host_scan.h"
#include "host_pipeline.h"
#include "host_util.h"
#include "host_grm.h"
