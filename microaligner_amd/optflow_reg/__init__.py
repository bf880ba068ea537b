from .flow_calc import TileFlowCalc, farneback
from .optflow_registrator import OptFlowRegistrator, merge_two_flows
from .warper import Warper
