cd $GRAFT_REPO_ROOT
bash tools/campaign.sh sweep "2 3 4" "X=jfilter" "VS_NO_JFILTER=1"
