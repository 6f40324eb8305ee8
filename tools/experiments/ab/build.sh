#!/bin/bash
# libA.so = the library at HEAD, libB.so = the library of the working tree (both under this folder).
set -e
cd "$(dirname "$0")/../../.."
make -s -C vings_mono_amd/csrc > /dev/null
cp vings_mono_amd/libvings_hip.so tools/experiments/ab/libB.so
git stash -q
make -s -C vings_mono_amd/csrc > /dev/null
cp vings_mono_amd/libvings_hip.so tools/experiments/ab/libA.so
git stash pop -q
make -s -C vings_mono_amd/csrc > /dev/null
echo built
