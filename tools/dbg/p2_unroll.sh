for f in tools/ubench/ubench_*; do echo $f; timeout 60 $f | tail -1; done
