cd $GRAFT_REPO_ROOT
for i in 1 2; do python tools/bench_echelle_auto.py 2>&1 | grep Echelle; done
cp starfish_amd/models/echelle_model.py /tmp/new.py; cp gpurun_old_echelle.py.txt starfish_amd/models/echelle_model.py
for i in 1 2; do python tools/bench_echelle_auto.py 2>&1 | grep Echelle | sed 's/^/OLD /'; done
cp /tmp/new.py starfish_amd/models/echelle_model.py
