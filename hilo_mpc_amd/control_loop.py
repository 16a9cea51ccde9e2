"""`SimpleControlLoop` (hilo_mpc/modules/control_loop.py:41-431) for batches: controller step -> plant step -> observer step,
`steps` times, for every instance of the batch at once.

What the reference's `_run` does per iteration (control_loop.py:343-397): `u = controller.optimize(x0[states of the
controller's model], cp=p)`, `plant.simulate(u=u, p=p)`, `observer.estimate()` - the controller is fed the PLANT state, the
observer runs alongside.  Here the plant is advanced on the device: with the controller's shooting map when it is the
controller's own model (`NMPC.plant_step`), with `Model.step` for a plant model of its own (a continuous plant is integrated with
eight classic Runge-Kutta steps per interval in place of the reference's CVODES), or by a callable `(x, u, p) -> x+`."""
import numpy as np
import torch


class SimpleControlLoop:
    def __init__(self, plant, controller, observer=None):
        if not hasattr(controller, 'optimize'):
            raise TypeError("the controller must offer optimize() (NMPC / LMPC)")
        self._controller, self._observer = controller, observer
        if callable(plant) and not hasattr(plant, 'dynamical_state_names'):
            self._plant_fun, self._plant = plant, None
        else:
            cm = getattr(controller, '_model', None)
            if plant is cm and hasattr(controller, 'plant_step'):
                self._plant_fun, self._plant = None, plant          # the controller's own shooting map
            else:
                # a plant of its own (other parameters, discretisation or equations than the controller's model): advanced
                # on the device by `Model.step` (hilo_pf_function with one particle per instance, no noise)
                if not hasattr(plant, 'step'):
                    raise TypeError("the plant must be a Model or a callable (x, u, p) -> x_next")
                if len(plant.input_names) != len(getattr(cm, 'input_names', plant.input_names)):
                    raise ValueError("plant and controller model have different numbers of inputs")
                self._plant_fun, self._plant = (lambda x, u, p: plant.step(x, u, p)[0]), plant
        self.solution = None

    def _measure(self, x):
        obs = self._observer
        names = list(getattr(obs._model, 'measurement_names', []))
        states = list(obs._model.dynamical_state_names)
        mnames = getattr(obs, '_measured_states', None)
        if mnames is None:
            raise NotImplementedError("pass `measure=` to run(): a callable x -> y for the observer")
        return x[:, [states.index(n) for n in mnames]]

    def run(self, steps, x0, p=None, measure=None, **kwargs):
        """x0 [B, nx] (numpy or device tensor).  Returns (and stores as `.solution`) a dict with the closed-loop trajectories
        x [steps + 1, B, nx], u [steps, B, nu], the solver status per step and, with an observer, its estimates."""
        c = self._controller
        tensor = isinstance(x0, torch.Tensor)
        x = x0 if tensor else np.atleast_2d(np.asarray(x0, dtype=float))
        X, U, S, E = [x], [], [], []
        for _ in range(int(steps)):
            u = c.optimize(x, cp=p, **kwargs) if p is not None else c.optimize(x, **kwargs)       # control_loop.py:362
            st = getattr(c, 'solver_status_code', None)
            S.append(None if st is None else np.asarray(st).copy())
            if self._plant_fun is not None:
                x = self._plant_fun(x, u, p)                                                        # control_loop.py:385
            else:
                xn = c.plant_step(x, u, cp=p)
                x = xn if tensor else xn.cpu().numpy()
            if self._observer is not None:                                                          # control_loop.py:388-397
                y = (measure or self._measure)(x)
                E.append(self._observer.estimate(y=y, u=u))
            X.append(x)
            U.append(u)
        stack = (lambda a: torch.stack(list(a))) if tensor else (lambda a: np.stack([np.asarray(q) for q in a]))
        self.solution = {'x': stack(X), 'u': stack([torch.as_tensor(q, device=x.device) if tensor and not isinstance(q, torch.Tensor)
                                                    else q for q in U]),
                         'status': S, 'estimates': E}
        return self.solution
