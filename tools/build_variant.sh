#!/bin/bash
# developer aid: a tagged build of the library beside the product's (lib/libvgan_gpu_<tag>.so) with extra -D flags
#   tools/build_variant.sh <tag> [-DWV_OCC=5 ...]
tag=$1; shift
VGAN_BUILD_TAG=_$tag VGAN_EXTRA_FLAGS="$*" python3 -m vgan_amd.build >/dev/null && echo "built vgan_amd/lib/libvgan_gpu_$tag.so ($*)"
