"""CPU oracle for the hilo_mpc_amd hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in ``hilo_mpc_amd/`` may import this package; only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do.

Each module is a plain numpy/sympy restatement of one slice of the reference hot path
(HILO-MPC v1.1.0), citing the reference ``file:line`` it follows:

* ``oracle.models``  - model zoo as sympy expressions + ERK discretisation
  (``hilo_mpc/util/modeling.py:1008-1085,1213-1281``, ``hilo_mpc/library/models.py``)
* ``oracle.kf``      - KF / EKF / UKF predict+update (``hilo_mpc/modules/estimator/kf.py:71-307,486-604``)
* ``oracle.gp``      - kernels, means, exact GP inference
  (``hilo_mpc/modules/machine_learning/gp/{kernel,mean,inference,gp}.py``)
* ``oracle.nmpc``    - multiple-shooting transcription (``hilo_mpc/modules/controller/mpc.py:1133-1787``)
  + dense primal-dual interior point (IPOPT lives in the un-vendored ``casadi`` dependency)
* ``oracle.lmpc``    - LMPC QP (``mpc.py:2143-2394``)
* ``oracle.smpc``    - deterministic surrogate of the stochastic NMPC (``mpc.py:2512-2645``)
* ``oracle.pf``      - particle filter (``hilo_mpc/modules/estimator/pf.py``)
* ``oracle.mhe``     - MHE transcription (``hilo_mpc/modules/estimator/mhe.py:418-790``)

Pinning status (see DESIGN.md section "Oracle"):
  KF / EKF / UKF / kernels / means / GP-LML : pinned by the reference's own known-answer tests
  NMPC interior point + collocation          : pinned by the numbers the reference's CSTR notebook prints
                                               (tests/golden/nmpc_cstr.json)
  multiple-shooting NMPC / LMPC / MHE / SMPC / particle filter : PARITY UNPINNED (the reference holds no numeric
                                               assertion for them and CasADi/IPOPT is not installable
                                               here); cross-checked by an independent scipy solver.
  oracle/cpu                                 : the C++/OpenMP CPU baseline of bench.py, validated against oracle/nmpc.py
"""
