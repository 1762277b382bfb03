import importlib,sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
synth=importlib.import_module("network-aware-bwa_amd.synth")
for tb in (2<<30,):
  for bpa,ch in ((16,1),(16,4),(32,1),(32,2),(64,1),(64,2),(64,4),(128,1),(128,2)):
    for nb in (1024,2048,4096):
      g,m=synth.gather_ceiling(tb,bpa,ch,nb,1000)
      print("table %.1fGB bytes/access %3d chains %d blocks %4d (lanes %7d): %8.1f GB/s %9.1f Macc/s"%(tb/2**30,bpa,ch,nb,nb*256,g,m),flush=True)
