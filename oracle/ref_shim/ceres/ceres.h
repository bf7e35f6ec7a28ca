// oracle/ref_shim/ceres/ceres.h -- limap/base/infinite_line.h includes <ceres/ceres.h> but the files compiled
// into oracle/_ref use nothing of it (TEST INFRASTRUCTURE).
#pragma once
