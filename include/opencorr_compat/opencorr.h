// opencorr.h -- umbrella header of the MI355X drop-in for OpenCorr's FFTCC -> ICGN path
// (the reference's umbrella is src/opencorr.h:20-42; only the hot-path classes exist here).
#pragma once
#include "oc_deformation.h"
#include "oc_engines.h"
#include "oc_io.h"
#include "oc_types.h"
