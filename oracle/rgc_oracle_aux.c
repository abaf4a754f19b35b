/* placeholder: front-end / de-skew / fusion restatements are added here */
#include "rgc_oracle.h"
