/* stands in for the CUDA toolkit header of that name when the reference sources are compiled for oracle/_ref */
#include "kt_cuda_emul.h"
