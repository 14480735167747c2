bash tools/experiments/run_claim_pb.sh "" base small main || exit 1
bash tools/experiments/run_claim_pb.sh "--workload street" base small main || exit 1
bash tools/experiments/run_claim_pb.sh "--workload street_hard" base small main || exit 1
