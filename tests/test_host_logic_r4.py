"""Host-side rules added in round 4 (no GPU): the deferred-gradient switch, the conv schedule choice at the 8192-product
boundary, the process-level hardware-queue setting."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _py(code, env=None):
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run([sys.executable, '-c', 'import sys; sys.path.insert(0, %r)\n' % ROOT + code], capture_output=True, text=True,
                          timeout=300, env=e)


def test_deferred_joins_are_off_until_a_trainer_enables_them():
    r = _py('from u2mkd_amd import deferred\nprint(deferred.enabled())\ndeferred.enable()\nprint(deferred.enabled())\n'
            'deferred.enable(False)\nprint(deferred.enabled(), deferred.pending())')
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.split() == ['False', 'True', 'False', '[]']


def test_hardware_queue_setting_is_the_launchers_not_the_imports():
    """importing the package leaves the environment alone; distributed.configure_runtime() (called by bench.py /
    run_training.py before their first GPU call) sets GPU_MAX_HW_QUEUES=8 for a rank of a multi-rank job (WORLD_SIZE > 1 or
    U2MKD_FORCE_DDP=1) unless the user exported a value, and leaves the runtime's default (4) to a single-rank process, whose
    trainer then queues the next batch's geometry in slices (train._staged_geometry follows the same setting)."""
    base = {k: v for k, v in os.environ.items() if k not in ('GPU_MAX_HW_QUEUES', 'WORLD_SIZE', 'U2MKD_FORCE_DDP', 'U2MKD_STAGED_GEOMETRY')}
    code = ('import sys, os; sys.path.insert(0, %r); import u2mkd_amd; print(os.environ.get("GPU_MAX_HW_QUEUES"));'
            'from u2mkd_amd import distributed as D, train as T; D.configure_runtime(); print(os.environ.get("GPU_MAX_HW_QUEUES"), T._staged_geometry())' % ROOT)

    def run(extra):
        r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300, env=dict(base, **extra))
        assert r.returncode == 0, r.stderr[-1000:]
        return r.stdout.split()
    assert run({}) == ['None', 'None', 'True']
    assert run({'WORLD_SIZE': '8'}) == ['None', '8', 'False']
    assert run({'U2MKD_FORCE_DDP': '1'}) == ['None', '8', 'False']
    assert run({'WORLD_SIZE': '8', 'GPU_MAX_HW_QUEUES': '4'}) == ['4', '4', 'True']
    assert run({'GPU_MAX_HW_QUEUES': '8'}) == ['8', '8', 'False']
    assert run({'U2MKD_STAGED_GEOMETRY': '0'}) == ['None', 'None', 'False']


def test_schedule_choice_at_the_boundary():
    from u2mkd_amd.torchsparse.nn import functional as F
    old = os.environ.pop('U2MKD_CONV_SCHEDULE', None)
    try:
        assert not F._pairs_mode(64, 64, 80000) and not F._pairs_mode(96, 64, 80000)
        assert F._pairs_mode(96, 96, 80000) and F._pairs_mode(512, 512, 100)
        assert not F._pairs_mode(64, 128, 80000) and not F._pairs_mode(128, 64, 54000)      # the boundary: tiles on large sets
        assert F._pairs_mode(64, 128, 16000) and F._pairs_mode(128, 64, 6000)                # ... pairs on small ones
        assert not F._pairs_mode(512, 510, 80000)                                            # columns not a multiple of 4
        os.environ['U2MKD_CONV_SCHEDULE'] = 'pairs'
        assert F._pairs_mode(32, 32, 10)
        os.environ['U2MKD_CONV_SCHEDULE'] = 'tiles'
        assert not F._pairs_mode(512, 512, 10)
    finally:
        os.environ.pop('U2MKD_CONV_SCHEDULE', None)
        if old is not None:
            os.environ['U2MKD_CONV_SCHEDULE'] = old
