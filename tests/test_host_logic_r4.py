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
    run_training.py before their first GPU call) sets GPU_MAX_HW_QUEUES=8 unless the user exported a value."""
    env = {k: v for k, v in os.environ.items() if k != 'GPU_MAX_HW_QUEUES'}
    r = subprocess.run([sys.executable, '-c', 'import sys, os; sys.path.insert(0, %r); import u2mkd_amd; print(os.environ.get("GPU_MAX_HW_QUEUES"));'
                        'from u2mkd_amd import distributed as D; D.configure_runtime(); print(os.environ.get("GPU_MAX_HW_QUEUES"))' % ROOT],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and r.stdout.split() == ['None', '8'], (r.stdout, r.stderr[-1000:])
    r = _py('import os, u2mkd_amd\nfrom u2mkd_amd import distributed as D\nD.configure_runtime()\nprint(os.environ["GPU_MAX_HW_QUEUES"])', {'GPU_MAX_HW_QUEUES': '4'})
    assert r.returncode == 0 and r.stdout.strip() == '4'


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
