import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.build()
nets = importlib.import_module("prior-diffuse_amd.nets")
synth = importlib.import_module("prior-diffuse_amd.synth")
net = nets.EpsNetPlan(nets.Ctx("cuda:0"), synth.make_state_dict("DiffUNet1"), 32, 401, time_cond=True, nsteps=1, split_bf16=True, planes=int(os.environ.get("PLANES", "3")))
net.build_time(); net.build_step(0); net.finish()
net.x.normal_(); net.x_init.normal_(); net.tsteps.fill_(10.45)
st = torch.cuda.current_stream().cuda_stream
net.plan.run(st); torch.cuda.synchronize()
