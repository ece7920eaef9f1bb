for n in 200 512 2048; do
  echo "== $n jobs: default"; EIGEN_JOBS=$n FORMS=quad bash tools/eigen_ab.sh | grep launches
  for v in f2prio1 f2prio3; do echo "== $n jobs: $v"; EIGEN_JOBS=$n PCSF_EIGEN_FORM=2 FORMS=none bash tools/eigen_ab.sh tools/_build/lib_$v.so | grep launches; done
done
