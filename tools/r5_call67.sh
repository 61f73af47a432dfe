#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/r5_sixth.sh
bash tools/r5_seventh.sh
