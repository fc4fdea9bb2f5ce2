// BlobTree field / polygonizer C-ABI -- filled in below.
#include "common.h"
